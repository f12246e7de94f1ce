#!/bin/bash
# PMC passes over one dense tile-GEMM launch pair: tools/pmc_gemm.sh <f32|f64> <outdir>
# (separate rocprofv3 --pmc passes; kernel-trace only, as the GPU pool requires)
dt=${1:-f32}; out=${2:-gpurun_out/pmc_gemm}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 tools/exp_one_gemm.py 8192 8192 8192 $dt 2 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; }
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.OrderedDict()
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "gemm_kernel" not in r["Kernel_Name"]:
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, v in acc.items():
        line = f"{k:32s} {sum(v)/len(v):.4e}  (n={len(v)})"
        print(line); fo.write(line + "\n")
PY
rm -rf $out/p[0-9]
