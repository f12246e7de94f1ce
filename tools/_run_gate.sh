cd /root/repo
(echo "# tools/exp_fp32_gate.py, round 4: the same 41 models as r03_fp32_gate_calibration.log, batches of 131 072 and (N = 65 536 model) 1 048 576 queries"; 
 echo "## QUERIES=131072 CASES=40"; QUERIES=131072 CASES=40 timeout -k 10 500 python tools/exp_fp32_gate.py 2>&1 | grep "^N=";
 echo "## QUERIES=1048576 CASES=0 --big"; QUERIES=1048576 CASES=0 timeout -k 10 300 python tools/exp_fp32_gate.py --big 2>&1 | grep "^N=") > gpurun_out/r04_fp32_gate_calibration.log
tail -5 gpurun_out/r04_fp32_gate_calibration.log
python - <<'PY'
import re
v=[];m=[]
for l in open('gpurun_out/r04_fp32_gate_calibration.log'):
    g=re.search(r"err/A2q valu ([0-9.e+-]+|nan) mfma ([0-9.e+-]+|nan)", l)
    a=re.search(r"A2 train\s+([0-9.]+) query\s+([0-9.]+)", l)
    if g and a and float(a.group(2))>=10:
        v.append(float(g.group(1)));
        if g.group(2)!='nan': m.append(float(g.group(2)))
print("max err/A2q (A2q>=10): valu", max(v), "mfma", max(m), "n", len(v), len(m))
PY
