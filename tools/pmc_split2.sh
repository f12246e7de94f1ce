#!/bin/bash
# clock / MFMA-busy / LDS / wait counters of the fp16 x 2 variance launch at the headline shape (tools/pmc_split2.sh [outdir])
out=${1:-gpurun_out/pmc_split2}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  FORMS=1 SPLIT2_TILES=0,1 REPS=2 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 tools/exp_k5_forms.py > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
done
python3 - $out <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
for pat, name in (("k5_split_kernel<4, 2, 4>", "fp16x2_512x128"), ("k5_split_kernel<4, 2, 2>", "fp16x2_256x128"), ("k5_split_kernel<4, 3, 2>", "bf16x3")):
    acc = collections.OrderedDict(); dur = []
    for f in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
    with open(f"{d}/pmc_mfma_k5_{name}.txt", "w") as o:
        o.write(f"kernel filter: {pat}\n")
        for k, v in acc.items():
            o.write(f"{k:32s} {sum(v)/len(v):.4e}  (n={len(v)})\n")
        if dur:
            t = sum(dur) / len(dur); g = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"])
            mf = sum(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(acc["SQ_VALU_MFMA_BUSY_CYCLES"])
            o.write(f"derived: duration {t:.2f} ms; clock = GRBM_GUI_ACTIVE / 8 / duration = {g / 8 / (t * 1e-3) / 1e9:.3f} GHz; "
                    f"MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) = {mf / (g / 8 * 1024):.3f}\n")
    print(open(f"{d}/pmc_mfma_k5_{name}.txt").read())
PY
rm -rf $out/p[0-9]
