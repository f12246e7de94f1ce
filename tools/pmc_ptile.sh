#!/bin/bash
# Counter passes over the one-launch Cholesky (tools/pmc_ptile.sh N [outdir]): matrix-pipe busy, clock, wave waits, L2 hit rate,
# fabric bytes.  One rocprofv3 --pmc run per counter set (never combined with the trace domains gpurun refuses).
n=${1:-16384}; out=${2:-gpurun_out/pmc_ptile}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 tools/exp_potrf_trace.py $n > $out/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $out/p$i.log; }
done
python3 - $out $n <<'PY'
import csv, glob, sys, collections
d, n = sys.argv[1], int(sys.argv[2])
acc = collections.OrderedDict(); dur = []
for f in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "ptile_potrf_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
with open(f"{d}/pmc_ptile_{n}.txt", "w") as o:
    o.write(f"ptile_potrf_kernel, N = {n} (tools/exp_potrf_trace.py under rocprofv3 --pmc, one pass per counter set)\n")
    for k, v in acc.items():
        o.write(f"{k:32s} {sum(v)/len(v):.4e}  (n={len(v)})\n")
    if dur and "SQ_VALU_MFMA_BUSY_CYCLES" in acc:
        t = sum(dur) / len(dur); g = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"])
        mf = sum(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(acc["SQ_VALU_MFMA_BUSY_CYCLES"])
        o.write(f"derived: duration {t:.3f} ms = {n ** 3 / 3 / (t * 1e-3) / 1e12:.1f} TF; clock = GRBM_GUI_ACTIVE / 8 / duration = {g / 8 / (t * 1e-3) / 1e9:.3f} GHz; "
                f"MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) = {mf / (g / 8 * 1024):.3f}\n")
    if "TCC_HIT_sum" in acc:
        hit = sum(acc["TCC_HIT_sum"]) / len(acc["TCC_HIT_sum"]); miss = sum(acc["TCC_MISS_sum"]) / len(acc["TCC_MISS_sum"])
        o.write(f"derived: L2 hit rate {hit / (hit + miss):.3f}\n")
    if "FETCH_SIZE" in acc and "WRITE_SIZE" in acc:
        fs = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"]); ws = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"])
        o.write(f"derived: fabric bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB = {(2 * fs + ws) * 1024:.3e} (matrix: {n * n * 8:.3e} bytes)\n")
print(open(f"{d}/pmc_ptile_{n}.txt").read())
PY
rm -rf $out/p[0-9]
