#!/bin/bash
# PMC passes over the split-K5 launch: tools/pmc_k5split.sh <outdir>
out=${1:-gpurun_out/pmc_k5s}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  EXP_N=32768 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 tools/exp_k5_split.py > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.OrderedDict()
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "k5_split_kernel" not in r["Kernel_Name"]:
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:32s} {sum(v)/len(v):.4e}  (n={len(v)})")
PY
rm -rf $out/p[0-9]
