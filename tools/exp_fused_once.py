#!/usr/bin/env python3
"""Three LML + gradient evaluations (gpk_lml_eval: up to 4608 rows the factor and the inverse factor are ONE launch of
ptile_potrf_kernel) - the program tools/pmc_ptile.sh runs under the counters for the fused launch.    python tools/exp_fused_once.py [N]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    be = get_backend(0)
    rng = np.random.default_rng(0)
    dev = DeviceGP(rng.standard_normal((n, 9)), rng.standard_normal((n, 1)), be)
    for _ in range(3):
        ld, quad, g = dev.lml_eval(2.0, 1.0, 0.1001, 0.1, True)
    print(f"N={n}: logdet term {ld:.6f}, |grad| {np.abs(g).max():.3e}")


if __name__ == "__main__":
    main()
