#!/bin/bash
# Counter passes over the matrix-pipe mean kernel at C4 (one rocprofv3 --pmc run per counter set); summary in gpurun_out/pmc_c4/pmc_c4.txt
out=gpurun_out/pmc_c4; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
done
python3 - $out <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.OrderedDict(); dur = []
for f in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "mean_bf16_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
with open(d + "/pmc_c4.txt", "w") as o:
    for k, v in acc.items():
        o.write(f"{k:32s} {sum(v)/len(v):.4e}  (n={len(v)})\n")
    if dur:
        t = sum(dur) / len(dur); g = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"])
        o.write(f"duration {t:.3f} ms clock {g / 8 / (t * 1e-3) / 1e9:.3f} GHz\n")
print(open(d + "/pmc_c4.txt").read())
PY
rm -rf $out/p[0-9]
