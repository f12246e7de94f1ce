cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/exp_k3.py 512 1024 2048 4096 8192 2>&1 | grep N= | tee gpurun_out/k3_ab.log
