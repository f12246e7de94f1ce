cd /root/repo
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "default rc=$?"; tail -c 1500 gpurun_out/r04_bench_default.json
timeout -k 10 300 python bench.py --workload lml > gpurun_out/r04_bench_lml.json 2> gpurun_out/r04_bench_lml.err; echo "lml rc=$?"; cat gpurun_out/r04_bench_lml.json
timeout -k 10 400 python bench.py --workload train > gpurun_out/r04_bench_train.json 2> gpurun_out/r04_bench_train.err; echo "train rc=$?"; cat gpurun_out/r04_bench_train.json
