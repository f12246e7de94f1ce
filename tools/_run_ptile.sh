cd /root/repo
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
