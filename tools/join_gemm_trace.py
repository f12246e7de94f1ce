#!/usr/bin/env python3
"""Join the gemm_log lines (GPK_OPTS=gemm_log=1) (stderr, launch order) of tools/exp_potrf_trace.py with the rocprofv3 kernel
trace: time and TFLOP/s per class of GEMM launch.  usage: join_gemm_trace.py <stderr log> <kernel_trace.csv>"""
import collections
import csv
import sys

log, trace = sys.argv[1:3]
calls = []
phase = "pre"
for ln in open(log):
    if ln.startswith("POTRF_BEGIN"): phase = "potrf"
    elif ln.startswith("POTRF_END"): phase = "post"
    elif ln.startswith("GPKGEMM"):
        f = ln.split()
        calls.append((phase, int(f[1]), int(f[2]), int(f[3]), int(f[4]), f[5], f[6], f[7], [int(x) for x in f[9:12]], [int(x) for x in f[13:16]]))
rows = [r for r in csv.DictReader(open(trace)) if "gemm_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
assert len(rows) == len(calls), (len(rows), len(calls))
leaf = [r for r in csv.DictReader(open(trace)) if "leaf_kernel" in r["Kernel_Name"]]
agg = collections.OrderedDict()
for c, r in zip(calls, rows):
    phase, dt, m, n, k, ta, tb, lo, kb, ke = c
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    # flops with the per-tile k-ranges
    ntm, ntn = m // 128, n // 128
    fl = 0.0
    for tm in range(ntm):
        for tn in range(ntn):
            if lo == "lo1" and tn > tm: continue
            b = max(kb[0] + kb[1] * tm + kb[2] * tn, 0)
            e = k if ke[0] < 0 else min(ke[0] + ke[1] * tm + ke[2] * tn, k)
            fl += 2.0 * 128 * 128 * max(e - b, 0)
    key = (phase, f"{ta}{tb}{lo}", m, n, k)
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += dur; a[2] += fl
tot = collections.defaultdict(float)
print(f"{'phase':6s} {'kind':10s} {'m':>6s} {'n':>6s} {'k':>6s} {'calls':>6s} {'ms':>9s} {'TFLOP/s':>8s}")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{key[0]:6s} {key[1]:10s} {key[2]:6d} {key[3]:6d} {key[4]:6d} {a[0]:6d} {a[1]*1e3:9.3f} {a[2]/a[1]/1e12 if a[1] else 0:8.2f}")
    tot[key[0]] += a[1]
print("gemm time by phase (ms):", {k: round(v * 1e3, 2) for k, v in tot.items()})
print("leaf kernels:", len(leaf), "total ms", sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in leaf) * 1e-6)
first = min(int(r["Start_Timestamp"]) for c, r in zip(calls, rows) if c[0] == "potrf")
last = max(int(r["End_Timestamp"]) for c, r in zip(calls, rows) if c[0] == "potrf")
print("potrf span ms (first to last gemm):", (last - first) * 1e-6)
