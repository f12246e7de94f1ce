cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_ptile.py -x -q 2>&1 | tail -3
for r in 1 2; do echo "GPK_PTILE_PROG_ROWS=$r"; GPK_PTILE_PROG_ROWS=$r timeout -k 10 300 python tools/exp_ptile.py 512 1024 2048 4096 5120 2>&1 | grep N=; done
echo "GPK_PTILE_PROG_NT=128"; GPK_PTILE_PROG_NT=128 timeout -k 10 300 python tools/exp_ptile.py 8192 16384 2>&1 | grep N=
echo "default"; timeout -k 10 300 python tools/exp_ptile.py 8192 16384 2>&1 | grep N=
timeout -k 10 120 python tools/exp_ptile_trace.py 4096 2>&1 | grep -v amdgpu | head -14
