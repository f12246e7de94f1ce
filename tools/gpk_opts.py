"""Tools only: GPK_OPTS="name=value,name=value" (gpk_set_option names, include/gpk.h) applied to every libgpk handle the process
creates - the A/B switch of the experiment scripts (e.g. GPK_OPTS=ptile_inv_max_np=0 python tools/exp_lml_host.py 4096).
The library and the package themselves read no environment knobs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse(spec=None):
    spec = os.environ.get("GPK_OPTS", "") if spec is None else spec
    out = {}
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        out[k] = int(v) if v.lstrip("-").isdigit() else v
    return out


def install(spec=None):
    """Every Backend created from now on (and the shared ones that exist already) gets the options."""
    from unmanned_aerial_vehicles_amd import device
    opts = parse(spec)
    if not opts:
        return opts
    orig = device.Backend.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        self.set_options(**opts)

    device.Backend.__init__ = init
    for b in list(device._backends.values()) + [w[0] for pool in device._workers.values() for w in pool]:
        b.set_options(**opts)
    return opts
