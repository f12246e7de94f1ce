#!/bin/bash
# Profile set from ONE GPU box (tools/pmc_profiles.sh [outdir]); everything lands under gpurun_out/prof2/ and the summaries
# are copied to profiles/ (r03_*) by hand afterwards.
#   1. rocprofv3 --kernel-trace --stats of the default bench command         -> kernel_stats.csv + the bench line
#   2. FETCH_SIZE / WRITE_SIZE passes of the same command (separate passes)   -> pmc_hbm_traffic.md, pmc_traffic.json
#   3. MFMA-busy / clock / wait / L2 counters of the variance launches (fp16 x 2, bf16 x 3, fp32 MFMA): tools/pmc_k5.sh
# (counter passes use --kernel-trace only, as the GPU pool requires; the program after "--" is python3 itself)
out=${1:-gpurun_out/prof2}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- $B --steps 10 --warmup 3 > $out/bench_under_rocprof.json 2> $out/prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv && echo "kernel stats ok"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -- $B --steps 2 --warmup 0 > $out/pmc_$c.json 2> $out/pmc_$c.err
done
python3 - $out <<'PY'
import csv, glob, json, sys, collections, subprocess
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    mx = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]; v = float(r["Counter_Value"]); mx[k] = max(mx[k], v); n[k] += 1
    res[c] = (mx, n)
names = sorted(res["FETCH_SIZE"][0], key=lambda k: -res["FETCH_SIZE"][0][k])[:16]
with open(out + "/pmc_hbm_traffic.md", "w") as fo:
    fo.write("`rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (two separate passes) over "
             "`python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 0`; values in KB as rocprofv3 reports them; "
             "corrected bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE counts 64 B per 128-B request on "
             "16-byte-per-lane streams: MI355X_MICROARCH.md, HBM section).\n\n")
    fo.write("| kernel | calls | FETCH_SIZE max/launch (KB) | WRITE_SIZE max/launch (KB) | corrected bytes, largest launch |\n|---|---|---|---|---|\n")
    for k in names:
        f_, w_ = res["FETCH_SIZE"][0][k], res["WRITE_SIZE"][0].get(k, 0.0)
        fo.write(f"| `{k[:110]}` | {res['FETCH_SIZE'][1][k]} | {f_:.0f} | {w_:.0f} | {(2 * f_ + w_) * 1024:.3e} |\n")
entries = []
for key, pat in (("k5_direct_kernel", "k5_direct_kernel<4>"), ("k5_split_kernel", "k5_split_kernel<4>")):
    ks = [k for k in res["FETCH_SIZE"][0] if pat in k]
    if ks:
        k = ks[0]
        entries.append({"kernel_key": key, "kernel": k, "n_train": 65536, "queries": 10000,
                        "bytes_per_launch": (2 * res["FETCH_SIZE"][0][k] + res["WRITE_SIZE"][0].get(k, 0.0)) * 1024,
                        "fetch_size_kb": res["FETCH_SIZE"][0][k], "write_size_kb": res["WRITE_SIZE"][0].get(k, 0.0),
                        "source": "profiles/r03_pmc_hbm_traffic.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py; "
                                  "(2 x FETCH_SIZE + WRITE_SIZE) x 1024)"})
json.dump({"entries": entries}, open(out + "/pmc_traffic.json", "w"), indent=1)
print(open(out + "/pmc_hbm_traffic.md").read())
PY
rm -rf $out/prof $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
[ "$ONLY_TRAFFIC" = "1" ] && exit 0      # steps 1 and 2 only
# 3. clock / MFMA busy / waits / L2 hit rate of the variance launches (tools/pmc_k5.sh -> pmc_k5_<form>.txt)
bash tools/pmc_k5.sh $out/k5 split2,bf16x3,fp32
cp $out/k5/pmc_k5_*.txt $out/ 2>/dev/null
