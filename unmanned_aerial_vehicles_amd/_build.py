"""Builds libgpk.so (the HIP kernels + C ABI) in-tree for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
`unmanned_aerial_vehicles_amd/libgpk.so` travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libgpk.so")
SOURCES = ["gpk_api.hip", "gpk_gram.hip", "gpk_gemm.hip", "gpk_chol.hip", "gpk_ptile.hip", "gpk_grad.hip", "gpk_mean.hip", "gpk_k5split.hip", "gpk_small.hip",
           "gpk_model.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(os.path.dirname(PKG_DIR), "include", "gpk.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), variant=None):
    """Compile every .hip translation unit and link libgpk.so.  Returns the library path.

    `variant` (with `extra_flags`, e.g. ("-DGPK_PIPE=2",)) builds an experimental copy
    build/libgpk_<variant>.so instead; select it at run time with GPK_LIBRARY=<path>."""
    if variant is None and not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libgpk.so")
    objdir = os.path.join(PKG_DIR, "build") if variant is None else os.path.join(PKG_DIR, "build", "var_" + variant)
    lib_path = LIB_PATH if variant is None else os.path.join(PKG_DIR, "build", f"libgpk_{variant}.so")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if verbose and out:
            print(out)
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    tmp = lib_path + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    os.replace(tmp, lib_path)
    return lib_path


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1:      # python -m unmanned_aerial_vehicles_amd._build <variant> [-DFLAG ...]
        print(build(variant=sys.argv[1], extra_flags=tuple(sys.argv[2:]), verbose=True))
    else:
        print(build(force=True, verbose=True))
