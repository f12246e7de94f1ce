#!/bin/bash
# Round artifacts from ONE GPU box: tests, smoke, default bench, rocprofv3 stats of the same command, the C4 and gram
# workloads, the launcher rehearsal, the BASELINE configuration table, serving latency.  Everything lands under
# gpurun_out/final/ (tools/pmc_profiles.sh produces the counter summaries separately).
set -o pipefail
out=gpurun_out/final; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json 2> $out/prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv; rm -rf $out/prof
python bench.py --workload c4 --steps 5 > $out/bench_c4.json 2> $out/bench_c4.err; echo "c4 rc=$?"
python bench.py --workload gram --steps 8 > $out/bench_gram.json 2> $out/bench_gram.err; echo "gram rc=$?"
# the distributed code path (process group, RCCL all-gather) with the one rank this box has: started by the driver's
# command line, and by bench.py's own launcher (parent -> torch.distributed.run -> rank)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/bench_torchrun_1rank.json 2> $out/bench_torchrun_1rank.err; echo "torchrun rc=$?"
BENCH_FORCE_LAUNCH=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/bench_via_launcher_1rank.json 2> $out/bench_via_launcher_1rank.err; echo "launcher rc=$?"
python -u tools/run_configs.py > $out/run_configs.log 2>&1; echo "run_configs rc=$?"; tail -3 $out/run_configs.log
timeout -k 10 200 python -u tools/exp_host_overhead.py 2>&1 | grep -v amdgpu.ids > $out/serving_latency.txt; echo "latency rc=$?"; head -8 $out/serving_latency.txt
# per-launch-class breakdown of one N = 65536 factorisation
GPK_GEMM_LOG=1 rocprofv3 --kernel-trace --output-format csv -d $out/ptrace -- python3 tools/exp_potrf_trace.py > $out/potrf_trace.out 2> $out/potrf_trace.err
python3 tools/join_gemm_trace.py $out/potrf_trace.err $(find $out/ptrace -name "*kernel_trace.csv" | head -1) > $out/potrf_launch_breakdown.txt 2>&1; tail -4 $out/potrf_launch_breakdown.txt; rm -rf $out/ptrace
(FUZZ_SEED=0 FUZZ_CASES=100 timeout -k 10 400 python -u tools/fuzz_parity.py) 2>&1 | grep -v amdgpu.ids > $out/fuzz_parity.log; echo "fuzz rc=$?"; grep "cases, " $out/fuzz_parity.log
