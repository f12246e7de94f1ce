#!/bin/bash
# leaf kernel duration per variant library: tools/exp_leaf_time.sh lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/leaf
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" != "default" ]; then export GPK_LIBRARY=$PWD/$lib; else unset GPK_LIBRARY; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/leaf/$name -- python3 tools/exp_potrf_trace.py 4096 > gpurun_out/leaf/$name.log 2>&1
  f=$(find gpurun_out/leaf/$name -name "*kernel_stats.csv" | head -1)
  python3 - "$name" "$f" <<'PY'
import csv, sys
name, f = sys.argv[1:3]
for r in csv.DictReader(open(f)):
    if "leaf_kernel" in r["Name"]:
        print(f"== {name}: leaf calls {r['Calls']} avg {float(r['AverageNs'])/1e3:.1f} us  min {float(r['MinNs'])/1e3:.1f} us")
PY
  rm -rf gpurun_out/leaf/$name
done
