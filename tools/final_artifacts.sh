#!/bin/bash
# Round-5 artifacts from the GPU box (everything lands under gpurun_out/final5/; what is judged is copied into profiles/ as r05_*).
# Four parts, one gpurun call each (a call is limited to 20 minutes):
#   bash tools/final_artifacts.sh p     counter passes: K5 under bench.py (FETCH / WRITE, MFMA busy, clock, L2), the one-launch
#                                       factorisation at three sizes and with the inverse factor's tiles
#   bash tools/final_artifacts.sh w     counter passes of the workloads lml / train / gram / c4 (fabric bytes per step).
#                                       THEN, here: python tools/merge_pmc.py r05 ... -> profiles/pmc_traffic.json (bench.py reads it)
#   bash tools/final_artifacts.sh a     smoke, bench lines (default, under rocprofv3 --stats, lml, train, c4, gram, torchrun with one rank)
#   bash tools/final_artifacts.sh b     A/B logs, traces, BASELINE configurations, latency table, fuzz, stress, pytest -m gpu
set -o pipefail
out=gpurun_out/final5; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
part=${1:-a}
stats() {   # stats NAME -- bench args: rocprofv3 --kernel-trace --stats of a bench command -> $out/NAME_kernel_stats.csv + the line
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py "$@" > $out/bench_${name}_under_rocprof.json 2> $out/prof_$name.err
  cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/${name}_kernel_stats.csv; rm -rf $out/prof
}
if [[ $part == *p* ]]; then
ONLY_TRAFFIC=1 bash tools/pmc_profiles.sh $out/prof2 > $out/pmc_profiles.log 2>&1; echo "pmc_profiles rc=$?"
bash tools/pmc_k5.sh $out/k5 split2 > $out/pmc_k5.log 2>&1; echo "pmc_k5 rc=$?"
for n in 4096 8192 16384; do bash tools/pmc_ptile.sh $n $out/pmc_ptile > $out/pmc_ptile_$n.log 2>&1; done
bash tools/pmc_ptile.sh 4096 $out/pmc_ptile fused > $out/pmc_ptile_fused.log 2>&1; echo "pmc_ptile done"
fi
if [[ $part == *w* ]]; then
for w in lml train gram c4; do bash tools/pmc_workload.sh $w $out/pmc_$w > $out/pmc_$w.log 2>&1; echo "pmc_workload $w rc=$?"; done
fi
if [[ $part == *a* ]]; then
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
stats bench --no-cpu-baseline --no-extras
python bench.py --workload lml > $out/bench_lml.json 2> $out/bench_lml.err; echo "lml rc=$?"
stats lml --workload lml --no-cpu-baseline
python bench.py --workload train > $out/bench_train.json 2> $out/bench_train.err; echo "train rc=$?"
stats train --workload train --no-cpu-baseline --steps 1 --warmup 1
python bench.py --workload c4 --steps 5 > $out/bench_c4.json 2> $out/bench_c4.err; echo "c4 rc=$?"
python bench.py --workload gram --steps 6 > $out/bench_gram.json 2> $out/bench_gram.err; echo "gram rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/bench_torchrun_1rank.json 2> $out/bench_torchrun_1rank.err; echo "torchrun rc=$?"
fi
if [[ $part == *b* ]]; then
(echo "# tools/exp_ptile.py: gpk_potrf by the recursive launch chain (ptile = 0) against the one-launch tile factorisation, same matrix, same box"; python tools/exp_ptile.py 512 1024 2048 4096 5120 8192 10112 16384 32768 2>&1 | grep N=) > $out/ptile_ab.log; cat $out/ptile_ab.log
(echo "# tools/exp_balanced.py: gpk_trtri and gpk_wtw with the static tile mapping (gemm_balanced = 0) against the balanced persistent schedule, same factor, same box"; python tools/exp_balanced.py 1024 2048 4096 8192 10112 16384 2>&1 | grep N= | cut -c1-260) > $out/gemm_balanced_ab.log
for cfg in "4096" "4096 batch" "1024"; do tag=$(echo $cfg | tr ' ' _); rm -rf /tmp/lt; rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -- python3 tools/exp_lml_trace.py $cfg > $out/lml_trace_$tag.log 2>&1; python3 tools/exp_lml_trace.py --join /tmp/lt >> $out/lml_trace_$tag.log 2>&1; done
python tools/exp_ptile_trace.py 1024 2>&1 | grep -v amdgpu > $out/ptile_trace_1024.log
python tools/exp_ptile_trace.py 4096 2>&1 | grep -v amdgpu > $out/ptile_trace_4096.log; head -8 $out/ptile_trace_4096.log
python -u tools/run_configs.py > $out/run_configs.log 2>&1; echo "run_configs rc=$?"
(echo "# tools/exp_lml_host.py: one LML + gradient evaluation (wall), factor + inverse factor as one launch against factor, then level-by-level inverse (GPK_OPTS=ptile_inv_max_np=0)"; for n in 1000 2048 3000 4096 5120; do python tools/exp_lml_host.py $n 2>&1 | grep N= | sed 's/$/  [one launch up to 4608 rows]/'; GPK_OPTS=ptile_inv_max_np=0 python tools/exp_lml_host.py $n 2>&1 | grep N= | sed 's/$/  [level by level]/'; done) > $out/lml_fused_ab.log
(echo "# tools/exp_c5_batch.py"; python tools/exp_c5_batch.py 4096 2>&1 | grep N=; python tools/exp_c5_batch.py 1000 2>&1 | grep N=) > $out/c5_batch.log
(echo "# tools/exp_train_small.py: fit with optimiser + 1 restart, P = 6"; for n in 1000 4096; do python tools/exp_train_small.py $n 1 2>&1 | tail -1; done) > $out/train_small.log
python tools/serving_latency.py 1000 4096 10000 2>&1 | grep -v amdgpu > $out/serving_latency.txt; tail -4 $out/serving_latency.txt
python tools/cpu_gpu_stage_table.py > $out/cpu_gpu_stages.log 2>&1; echo "stage table rc=$?"
python tools/fuzz_parity.py > $out/fuzz_parity.log 2>&1; tail -1 $out/fuzz_parity.log
python tools/stress_ptile.py 100 2>&1 | grep -v amdgpu > $out/stress_ptile.log; tail -1 $out/stress_ptile.log
python -m pytest tests -m gpu -q -rs 2>&1 | tail -6 > $out/pytest_gpu.log; tail -2 $out/pytest_gpu.log
fi
