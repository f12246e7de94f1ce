cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_ptile.py -x -q 2>&1 | tail -3
timeout -k 10 400 python tools/exp_ptile.py 4096 16384 32768 65536 2>&1 | grep N=
