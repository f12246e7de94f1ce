#!/bin/bash
# Counter passes over the variance launches at the headline shape (tools/pmc_k5.sh [outdir] [FORMS of tools/exp_k5_direct.py]):
# clock / MFMA busy, wave wait breakdown, LDS and vector-memory instruction counts, L2 hit rate, fabric bytes.
# One rocprofv3 --pmc run per counter set (never combined with the trace domains gpurun refuses).
out=${1:-gpurun_out/pmc_k5}; forms=${2:-split2,bf16x3}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  SWEEP=0 FORMS=$forms REPS=2 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 tools/exp_k5_direct.py > $out/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $out/p$i.log; }
done
python3 - $out <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
for pat, name in (("k5_direct_kernel<4>", "fp16x2_direct"), ("k5_split_kernel<4>", "bf16x3"), ("gemm_kernel<float, false, false, 1", "fp32_mfma")):
    acc = collections.OrderedDict(); dur = []
    for f in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
    if not acc:
        continue
    with open(f"{d}/pmc_k5_{name}.txt", "w") as o:
        o.write(f"kernel filter: {pat}\n")
        for k, v in acc.items():
            o.write(f"{k:32s} {sum(v)/len(v):.4e}  (n={len(v)})\n")
        if dur and "SQ_VALU_MFMA_BUSY_CYCLES" in acc:
            t = sum(dur) / len(dur); g = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"])
            mf = sum(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(acc["SQ_VALU_MFMA_BUSY_CYCLES"])
            o.write(f"derived: duration {t:.2f} ms; clock = GRBM_GUI_ACTIVE / 8 / duration = {g / 8 / (t * 1e-3) / 1e9:.3f} GHz; "
                    f"MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) = {mf / (g / 8 * 1024):.3f}\n")
        if "TCC_HIT_sum" in acc:
            hit = sum(acc["TCC_HIT_sum"]) / len(acc["TCC_HIT_sum"]); miss = sum(acc["TCC_MISS_sum"]) / len(acc["TCC_MISS_sum"])
            o.write(f"derived: L2 hit rate {hit / (hit + miss):.3f}\n")
        if "FETCH_SIZE" in acc:
            o.write(f"derived: fabric read bytes per launch = 2 x FETCH_SIZE x 1 KiB (gfx950 correction for 16-byte-per-lane "
                    f"streaming reads, MI355X_MICROARCH.md HBM section) = {2 * 1024 * sum(acc['FETCH_SIZE']) / len(acc['FETCH_SIZE']):.4e}\n")
    print(open(f"{d}/pmc_k5_{name}.txt").read())
PY
rm -rf $out/p[0-9]
