#!/bin/bash
# PMC passes over the K4 kernel: tools/pmc_mean.sh <valu|mfma> <outdir>
kern=${1:-mfma}; out=${2:-gpurun_out/pmc_mean}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 tools/exp_mean_one.py $kern > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; }
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.OrderedDict(); dur = []
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "mean_bf16_kernel" not in r["Kernel_Name"] and "mean_mfma_kernel" not in r["Kernel_Name"] and "predict_mean_kernel" not in r["Kernel_Name"]:
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for f in sorted(glob.glob(out + "/p1/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "mean_" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
with open(out + "/summary.txt", "w") as fo:
    for k, v in acc.items():
        line = f"{k:32s} {sum(v)/len(v):.4e}  (n={len(v)})"
        print(line); fo.write(line + "\n")
    print("durations ms", dur); fo.write(f"durations ms {dur}\n")
PY
rm -rf $out/p[0-9]
