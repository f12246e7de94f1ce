cd /root/repo
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04_pytest_gpu.log; cat gpurun_out/r04_pytest_gpu.log
