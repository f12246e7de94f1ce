#!/bin/bash
# Counter passes over the one-launch Cholesky (tools/pmc_ptile.sh N [outdir] [fused]): matrix-pipe busy, clock, wave waits, L2 hit rate,
# fabric bytes.  With "fused" the launch under the counters is the one with the inverse factor's tiles in its task list
# (tools/exp_fused_once.py: gpk_lml_eval with a gradient; 2 N^3 / 3 flops).  One rocprofv3 --pmc run per counter set (never combined with the trace domains gpurun refuses).
n=${1:-16384}; out=${2:-gpurun_out/pmc_ptile}; mkdir -p $out
prog=tools/exp_potrf_trace.py; tag=""; if [ "$3" == "fused" ]; then prog=tools/exp_fused_once.py; tag="_fused"; fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 $prog $n > $out/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $out/p$i.log; }
done
python3 - $out $n "$tag" <<'PY'
import csv, glob, sys, collections
d, n, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
flops = n ** 3 / 3 * (2 if tag else 1)
acc = collections.OrderedDict(); dur = []
for f in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "ptile_potrf_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
with open(f"{d}/pmc_ptile{tag}_{n}.txt", "w") as o:
    o.write(f"ptile_potrf_kernel{' with the tiles of the inverse factor (factor + inverse factor, 2 N^3 / 3 flops)' if tag else ''}, N = {n} "
            f"({'tools/exp_fused_once.py' if tag else 'tools/exp_potrf_trace.py'} under rocprofv3 --pmc, one pass per counter set)\n")
    for k, v in acc.items():
        o.write(f"{k:32s} {sum(v)/len(v):.4e}  (n={len(v)})\n")
    if dur and "SQ_VALU_MFMA_BUSY_CYCLES" in acc:
        t = sum(dur) / len(dur); g = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"])
        mf = sum(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(acc["SQ_VALU_MFMA_BUSY_CYCLES"])
        o.write(f"derived: duration {t:.3f} ms = {flops / (t * 1e-3) / 1e12:.1f} TF; clock = GRBM_GUI_ACTIVE / 8 / duration = {g / 8 / (t * 1e-3) / 1e9:.3f} GHz; "
                f"MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) = {mf / (g / 8 * 1024):.3f}\n")
    if "TCC_HIT_sum" in acc:
        hit = sum(acc["TCC_HIT_sum"]) / len(acc["TCC_HIT_sum"]); miss = sum(acc["TCC_MISS_sum"]) / len(acc["TCC_MISS_sum"])
        o.write(f"derived: L2 hit rate {hit / (hit + miss):.3f}\n")
    if "FETCH_SIZE" in acc and "WRITE_SIZE" in acc:
        fs = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"]); ws = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"])
        o.write(f"derived: fabric bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB = {(2 * fs + ws) * 1024:.3e} (matrix: {n * n * 8:.3e} bytes)\n")
print(open(f"{d}/pmc_ptile{tag}_{n}.txt").read())
PY
rm -rf $out/p[0-9]
