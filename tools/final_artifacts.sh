#!/bin/bash
# Round artifacts from ONE GPU box: tests, smoke, default bench, rocprofv3 stats + PMC traffic of the same command,
# the C4 workload, the BASELINE configuration table.  Everything lands under gpurun_out/final/.
set -o pipefail
out=gpurun_out/final; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_$c.json 2> $out/pmc_$c.err
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    mx = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]; v = float(r["Counter_Value"]); mx[k] = max(mx[k], v); n[k] += 1
    res[c] = (mx, n)
names = sorted(res["FETCH_SIZE"][0], key=lambda k: -res["FETCH_SIZE"][0][k])[:14]
with open(out + "/pmc_hbm_traffic.md", "w") as fo:
    fo.write("| kernel | calls | FETCH_SIZE max/launch (KB) | WRITE_SIZE max/launch (KB) | corrected bytes, largest launch (2 x FETCH + WRITE) x 1024 |\n|---|---|---|---|---|\n")
    for k in names:
        f_, w_ = res["FETCH_SIZE"][0][k], res["WRITE_SIZE"][0].get(k, 0.0)
        fo.write(f"| `{k[:120]}` | {res['FETCH_SIZE'][1][k]} | {f_:.0f} | {w_:.0f} | {(2 * f_ + w_) * 1024:.3e} |\n")
print(open(out + "/pmc_hbm_traffic.md").read())
PY
rm -rf $out/prof $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
python bench.py --workload c4 --steps 5 > $out/bench_c4.json 2> $out/bench_c4.err; echo "c4 rc=$?"
python -u tools/run_configs.py > $out/run_configs.log 2>&1; echo "run_configs rc=$?"; tail -3 $out/run_configs.log
# the distributed code path (process group, RCCL all-gather) with the one rank this box has
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_torchrun_1rank.json 2> $out/bench_torchrun_1rank.err; echo "torchrun rc=$?"
# control-loop latency (one row / horizon-25 through the estimator and the bare C call) and the randomised parity sweep
timeout -k 10 200 python -u tools/exp_host_overhead.py 2>&1 | grep -v amdgpu.ids > $out/serving_latency.txt; echo "latency rc=$?"; head -8 $out/serving_latency.txt
(FUZZ_SEED=0 FUZZ_CASES=150 timeout -k 10 400 python -u tools/fuzz_parity.py && FUZZ_SEED=1 FUZZ_CASES=100 timeout -k 10 400 python -u tools/fuzz_parity.py && FUZZ_SEED=2 FUZZ_CASES=50 FUZZ_MAX_N=7000 timeout -k 10 400 python -u tools/fuzz_parity.py) 2>&1 | grep -v amdgpu.ids > $out/fuzz_parity.log; echo "fuzz rc=$?"; grep "cases, " $out/fuzz_parity.log
