cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_ptile.py -x -q 2>&1 | tail -3
timeout -k 10 300 python tools/exp_ptile.py 512 1024 2048 4096 5120 8192 16384 2>&1 | grep N=
timeout -k 10 120 python tools/exp_ptile_trace.py 4096 2>&1 | grep -v amdgpu | head -14
