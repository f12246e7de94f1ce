"""The C ABI from a plain C caller: tests/c_abi/smoke.c is compiled with gcc against include/gpk.h, linked with
libgpk.so and the HIP runtime, and run as its own process (no Python, no torch in it).  It exercises K1-K5 through
hipMalloc'd buffers and checks size-independent properties plus the not-positive-definite return code."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_caller(tmp_path):
    gcc = shutil.which("gcc")
    rocm = "/opt/rocm"
    if gcc is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("gcc or the HIP headers are not available")
    pkg = os.path.join(ROOT, "unmanned_aerial_vehicles_amd")
    assert os.path.exists(os.path.join(pkg, "libgpk.so")), "libgpk.so is not built"
    exe = str(tmp_path / "c_abi_smoke")
    cmd = [gcc, "-O1", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", f"-I{ROOT}/include",
           os.path.join(ROOT, "tests", "c_abi", "smoke.c"), f"-L{pkg}", "-lgpk", f"-L{rocm}/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{pkg}", f"-Wl,-rpath,{rocm}/lib", "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    assert "C ABI smoke: OK" in r.stdout
