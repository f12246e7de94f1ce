cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_ptile.py -x -q 2>&1 | tail -5 > gpurun_out/ptile_test.log
cat gpurun_out/ptile_test.log
timeout -k 10 300 python tools/exp_ptile.py 1024 2048 4096 8192 16384 > gpurun_out/ptile_time.log 2>&1
cat gpurun_out/ptile_time.log
timeout -k 10 120 python tools/exp_ptile_trace.py 1024 > gpurun_out/ptile_trace_1024.log 2>&1
cat gpurun_out/ptile_trace_1024.log
