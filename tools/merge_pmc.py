#!/usr/bin/env python3
"""Build profiles/pmc_traffic.json (what bench.py reads `roofline.traffic` / `pmc` from) out of one round's counter passes:
    python tools/merge_pmc.py ROUND_TAG PROFILE_DIR K5_DIR [WORKLOAD_JSON ...]
PROFILE_DIR: output of tools/pmc_profiles.sh (pmc_traffic.json: FETCH / WRITE of the K5 launch under bench.py);
K5_DIR: output of tools/pmc_k5.sh (pmc_k5_fp16x2_direct.txt / _bf16x3.txt: MFMA busy, clock, L2 hit rate);
WORKLOAD_JSON: outputs of tools/pmc_workload.sh.  Sources are rewritten to the committed names profiles/<ROUND_TAG>_*."""
import json
import os
import re
import sys

tag, pdir, kdir = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
doc = json.load(open(os.path.join(pdir, "pmc_traffic.json")))
names = {"k5_direct_kernel": "fp16x2_direct", "k5_split_kernel": "bf16x3"}
for e in doc["entries"]:
    e["source"] = (f"profiles/{tag}_pmc_hbm_traffic.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py; "
                   "(2 x FETCH_SIZE + WRITE_SIZE) x 1024)")
    f = os.path.join(kdir, f"pmc_k5_{names.get(e['kernel_key'], 'x')}.txt")
    if os.path.exists(f):
        t = open(f).read()
        m = re.search(r"clock = [^=]*= ([0-9.]+) GHz; MFMA busy = [^=]*= ([0-9.]+)", t)
        h = re.search(r"L2 hit rate ([0-9.]+)", t)
        if m:
            e["pmc_clock_ghz"], e["pmc_mfma_busy"] = float(m.group(1)), float(m.group(2))
        if h:
            e["pmc_l2_hit_rate"] = float(h.group(1))
        e["pmc_source"] = (f"profiles/{tag}_pmc_k5_{names[e['kernel_key']]}.txt (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); "
                           "clock = GRBM_GUI_ACTIVE / 8 / duration; TCC_HIT / (TCC_HIT + TCC_MISS))")
    if e["kernel_key"] == "k5_direct_kernel":
        e["operand_bytes"] = 65536.0 * 65536 / 2 * 4 + 65536.0 * 10112 * 4      # lower tiles of W (4 B per entry) + K* in split form
doc["workloads"] = []
for wj in sys.argv[4:]:
    w = json.load(open(wj))
    w["source"] = w["source"].replace("profiles/r05_", f"profiles/{tag}_")
    doc["workloads"].append(w)
json.dump(doc, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
print("profiles/pmc_traffic.json:", [e["kernel_key"] for e in doc["entries"]], [w["workload"] for w in doc["workloads"]])
