cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/exp_balanced.py 512 1024 2048 4096 8192 10112 16384 32768 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/balanced_ab.log
