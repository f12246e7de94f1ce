#!/bin/bash
# Round-2 profile set from ONE GPU box (tools/pmc_profiles.sh [outdir]); everything lands under gpurun_out/prof2/ and
# the summaries are copied to profiles/ by hand afterwards.
#   1. rocprofv3 --kernel-trace --stats of the default bench command         -> kernel_stats.csv + the bench line
#   2. FETCH_SIZE / WRITE_SIZE passes of the same command (separate passes)   -> pmc_hbm_traffic.md, pmc_traffic.json
#   3. MFMA-busy / clock / LDS / wait counters for the three MFMA kernels north_star names: the bf16 x 3 split K5
#      launch (both forms), the fp32-MFMA K5 GEMM and the fp64 GEMM of the factorisation -> pmc_mfma_<what>.txt
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
for key, pat in (("k5_split2_kernel", "k5_split_kernel<4, 2, 4>"), ("k5_split_kernel", "k5_split_kernel<4, 3, 2>"),
                 ("k5_split16_kernel", "k5_split16_kernel")):
    ks = [k for k in res["FETCH_SIZE"][0] if pat in k]
    if ks:
        k = ks[0]
        entries.append({"kernel_key": key, "kernel": k, "n_train": 65536, "queries": 10000,
                        "bytes_per_launch": (2 * res["FETCH_SIZE"][0][k] + res["WRITE_SIZE"][0].get(k, 0.0)) * 1024,
                        "fetch_size_kb": res["FETCH_SIZE"][0][k], "write_size_kb": res["WRITE_SIZE"][0].get(k, 0.0),
                        "source": "profiles/r02_pmc_hbm_traffic.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py; "
                                  "(2 x FETCH_SIZE + WRITE_SIZE) x 1024)"})
json.dump({"entries": entries}, open(out + "/pmc_traffic.json", "w"), indent=1)
print(open(out + "/pmc_hbm_traffic.md").read())
PY
rm -rf $out/prof $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
[ "$ONLY_TRAFFIC" = "1" ] && exit 0      # steps 1 and 2 only
sets=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES" \
      "TCC_HIT_sum TCC_MISS_sum")
summ() {   # summ <dir> <kernel substring> <outfile>: average counters and the kernel's average duration
python3 - "$1" "$2" "$3" <<'PY'
import csv, glob, sys, collections
d, pat, fo = sys.argv[1:4]
acc = collections.OrderedDict(); dur = []
for f in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
with open(fo, "w") as o:
    o.write(f"kernel filter: {pat}   (averages over the launches of each pass; SQ_* wave counters are quad-cycles summed over waves,\n"
            f"SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CU_CYCLES are summed over the SIMDs / CUs, GRBM_GUI_ACTIVE over the 8 XCDs)\n")
    for k, v in acc.items():
        o.write(f"{k:32s} {sum(v)/len(v):.4e}  (n={len(v)})\n")
    if dur and "GRBM_GUI_ACTIVE" in acc and "SQ_VALU_MFMA_BUSY_CYCLES" in acc:
        t = sum(dur) / len(dur); g = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"])
        mf = sum(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(acc["SQ_VALU_MFMA_BUSY_CYCLES"])
        o.write(f"derived: duration under the counter pass {t:.2f} ms; clock = GRBM_GUI_ACTIVE / 8 / duration = {g / 8 / (t * 1e-3) / 1e9:.3f} GHz; "
                f"MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) = {mf / (g / 8 * 1024):.3f}\n")
print(open(fo).read())
PY
}
run_sets() {  # run_sets <dir> <program args...>
  d=$1; shift; mkdir -p $d; i=0
  for set in "${sets[@]}"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d/p$i -- "$@" > $d/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $d/p$i.log; }
  done
}
# split K5 (form 1 and form 2) at the headline shape
FORMS=1,2 REPS=2 run_sets $out/k5forms python3 tools/exp_k5_forms.py
summ $out/k5forms "k5_split_kernel<4, 3, 2>" $out/pmc_mfma_k5_split_form1.txt
summ $out/k5forms "k5_split16_kernel" $out/pmc_mfma_k5_split_form2.txt
summ $out/k5forms "k5_split_kernel<4, 2, 4>" $out/pmc_mfma_k5_fp16x2.txt
rm -rf $out/k5forms/p[0-9]
# fp32-MFMA K5 and the fp64 GEMMs of potrf / trtri: one bench run with --var-method inverse
run_sets $out/f32 python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 0 --var-method inverse
summ $out/f32 "gemm_kernel<float, false, false, 1" $out/pmc_mfma_k5_fp32.txt
summ $out/f32 "gemm_kernel<double, false, false, 0, 4, 128>" $out/pmc_mfma_potrf_gemm_f64.txt
rm -rf $out/f32/p[0-9]
