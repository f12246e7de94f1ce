#!/bin/bash
# Same-box A/B of libgpk variants of gpk_mean.hip on the C4 workload (variants: python -m unmanned_aerial_vehicles_amd._build <name>
# with the older gpk_mean.hip checked out; "head" = the in-tree library).  Results: profiles/r03_mean_direct_ab.log.
mkdir -p gpurun_out/ab
for v in oldmean directmean head oldmean head; do
  if [ $v = head ]; then lib=""; else lib=$PWD/unmanned_aerial_vehicles_amd/build/libgpk_$v.so; fi
  GPK_LIBRARY=$lib python bench.py --workload c4 --steps 10 > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err || { echo "$v failed"; tail -3 gpurun_out/ab/$v.err; }
  python -c "
import json
d=json.load(open('gpurun_out/ab/$v.json')); print('$v: %.3f ms per 2^20 queries, %.2f M pred/s, call-only %.3f ms, parity %.4e' % (d['ms_per_step'], d['value']/1e6, d['roofline']['k4_ms'], d['parity']['mean_max_rel_err_vs_fp64']))
" | tee -a gpurun_out/ab/ab.log
done
