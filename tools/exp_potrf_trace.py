#!/usr/bin/env python3
"""One potrf (+ optional trtri) at N for a per-launch breakdown: run under
   GPK_OPTS=gemm_log=1 rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/exp_potrf_trace.py N [trtri]
and join with tools/join_gemm_trace.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402
from tools import gpk_opts  # noqa: E402

gpk_opts.install()                # GPK_OPTS=ptile_xcd=1,... : A/B switches

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
be = get_backend(0)
rng = np.random.default_rng(0)
X = rng.standard_normal((N, 9))
dev = DeviceGP(X, np.zeros((N, 1)), be)
dev.gram(2.0, 1.0, 0.1001)
torch.cuda.synchronize()
print("POTRF_BEGIN", file=sys.stderr, flush=True)
try:
    dev.factorize(2.0, 1.0, 0.1001)
except Exception as e:  # noqa: BLE001  (skip-variant builds of the leaf kernel produce garbage)
    print("factorize:", e)
torch.cuda.synchronize()
print("POTRF_END", file=sys.stderr, flush=True)
if len(sys.argv) > 2:
    dev.inverse_factor(False)
    torch.cuda.synchronize()
print("done")
