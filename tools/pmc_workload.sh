#!/bin/bash
# Fabric/HBM bytes per STEP of a bench.py workload from rocprofv3 counter passes (tools/pmc_workload.sh WORKLOAD [outdir] [extra bench args]):
# FETCH_SIZE and WRITE_SIZE (separate passes, --kernel-trace only as the GPU pool requires) over
#     python3 bench.py --workload W --no-cpu-baseline --warmup 0 --steps S        for S = 1 and S = 3;
# bytes per step = (total at S = 3 - total at S = 1) / 2 - the set-up launches cancel -, corrected as MI355X_MICROARCH.md (HBM section)
# prescribes: (2 x FETCH_SIZE + WRITE_SIZE) x 1024.  A third pass (S = 1) reads SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE and
# TCC_HIT / TCC_MISS per kernel.  Writes <outdir>/pmc_<W>.md and <outdir>/pmc_<W>.json (an entry for profiles/pmc_traffic.json).
w=${1:-lml}; out=${2:-gpurun_out/pmc_$w}; shift 2; extra="$@"; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --workload $w --no-cpu-baseline --warmup 0 $extra"
for c in FETCH_SIZE WRITE_SIZE; do
  for s in 1 3; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/${c}_$s -- $B --steps $s > $out/${c}_$s.json 2> $out/${c}_$s.err || { echo "pass $c S=$s failed"; tail -3 $out/${c}_$s.err; }
  done
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/MFMA_1 -- $B --steps 1 > $out/MFMA_1.json 2> $out/MFMA_1.err || echo "pass MFMA failed"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/TCC_1 -- $B --steps 1 > $out/TCC_1.json 2> $out/TCC_1.err || echo "pass TCC failed"
python3 - $out $w "$extra" <<'PY'
import csv, glob, json, re, sys, collections
out, w, extra = sys.argv[1], sys.argv[2], sys.argv[3]
def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"^void ", "", k)
    return re.sub(r"\(.*$", "", k)[:90]
def totals(tag):
    acc, n = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(short(r["Kernel_Name"]), r["Counter_Name"])] += float(r["Counter_Value"]); n[(short(r["Kernel_Name"]), r["Counter_Name"])] += 1
    return acc, n
per = collections.defaultdict(lambda: [0.0, 0.0, 0.0])     # kernel -> [fetch KB / step, write KB / step, launches / step]
for ci, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    a1, n1 = totals(f"{c}_1"); a3, n3 = totals(f"{c}_3")
    for (k, cn), v in a3.items():
        per[k][ci] = (v - a1.get((k, cn), 0.0)) / 2.0
        per[k][2] = (n3[(k, cn)] - n1.get((k, cn), 0)) / 2.0
am, _ = totals("MFMA_1"); at, _ = totals("TCC_1")
rows = sorted(per.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1]))
step_bytes = sum((2 * f + wr) * 1024 for _, (f, wr, _) in rows)
line = json.loads(open(f"{out}/FETCH_SIZE_3.json").read().strip().splitlines()[-1])
with open(f"{out}/pmc_{w}.md", "w") as fo:
    fo.write(f"`rocprofv3 --kernel-trace --pmc <counter>` over `python3 bench.py --workload {w} --no-cpu-baseline --warmup 0 {extra} --steps S` "
             f"(one pass per counter; S = 1 and S = 3: per-step figures are (S = 3 minus S = 1) / 2, so the set-up launches cancel).  "
             f"Corrected bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024.  Workload: {line['config']['workload']}\n\n"
             f"**bytes per step: {step_bytes:.3e}**\n\n"
             "| kernel | launches / step | FETCH_SIZE KB / step | WRITE_SIZE KB / step | corrected bytes / step | MFMA busy | L2 hit |\n|---|---|---|---|---|---|---|\n")
    by = {}
    for k, (f, wr, nl) in rows[:14]:
        g = am.get((k, "GRBM_GUI_ACTIVE"), 0.0); mf = am.get((k, "SQ_VALU_MFMA_BUSY_CYCLES"), 0.0)
        hit, miss = at.get((k, "TCC_HIT_sum"), 0.0), at.get((k, "TCC_MISS_sum"), 0.0)
        busy = mf / (g / 8 * 1024) if g else None
        l2 = hit / (hit + miss) if hit + miss else None
        by[k] = {"launches_per_step": nl, "bytes_per_step": (2 * f + wr) * 1024, "mfma_busy": busy, "l2_hit_rate": l2}
        fo.write(f"| `{k}` | {nl:g} | {f:.0f} | {wr:.0f} | {(2 * f + wr) * 1024:.3e} | {'' if busy is None else f'{busy:.3f}'} | {'' if l2 is None else f'{l2:.3f}'} |\n")
json.dump({"workload": w, "n_train": line["config"].get("n_train") or int(re.search(r"N_train=(\d+)", line["config"]["workload"]).group(1)),
           "bytes_per_step": step_bytes, "by_kernel": by,
           "source": f"profiles/r05_pmc_{w}.md (tools/pmc_workload.sh {w}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --workload {w}, "
                     "S = 3 minus S = 1 over 2; (2 x FETCH_SIZE + WRITE_SIZE) x 1024)"}, open(f"{out}/pmc_{w}.json", "w"), indent=1)
print(open(f"{out}/pmc_{w}.md").read())
PY
rm -rf $out/FETCH_SIZE_[13] $out/WRITE_SIZE_[13] $out/MFMA_1 $out/TCC_1
