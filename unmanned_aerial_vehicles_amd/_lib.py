"""ctypes binding of libgpk.so (include/gpk.h).  No torch types cross this boundary:
device buffers are passed as integer addresses (`torch.Tensor.data_ptr()`), host arrays as
`numpy` double pointers.

The library is mandatory: if it is missing or cannot be loaded this module raises — there is
no CPU fallback for the product path.
"""
import ctypes as C
import os

import numpy as np

from ._build import LIB_PATH

GPK_F32, GPK_F64 = 0, 1
GPK_OK, GPK_NOT_PD, GPK_BAD_ARG, GPK_HIP_ERROR = 0, 1, 2, 3
GPK_TILE, GPK_MAX_D, GPK_MAX_P = 128, 64, 16
GPK_HOST_MAX_M = 4096
GPK_TIMED_K5, GPK_TIMED_GRAM, GPK_TIMED_GRAD, GPK_TIMED_POTRF = 1, 2, 3, 4

_vp, _i64, _int, _dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
_dp = C.POINTER(C.c_double)

# name -> (restype, argtypes); mirrors include/gpk.h one to one
SIGNATURES = {
    "gpk_create": (_int, [C.POINTER(_vp), _int]),
    "gpk_destroy": (None, [_vp]),
    "gpk_last_error": (C.c_char_p, [_vp]),
    "gpk_set_stream": (_int, [_vp, _vp]),
    "gpk_synchronize": (_int, [_vp]),
    "gpk_padded": (_i64, [_i64]),
    "gpk_fit": (_int, [_vp, _dp, _i64, _int, _dp, _int, _dp, _int, _dbl, _dbl, _dbl, _int]),
    "gpk_predict": (_int, [_vp, _vp, _i64, _vp, _vp, _int, _int]),
    "gpk_lml": (_int, [_vp, _dp, _int, _dp, _dp]),
    "gpk_fit_batched": (_int, [_vp, _int, _dp, _i64, _int, _dp, _dp, _int, _dp, _dp, _dbl, _int, C.POINTER(C.c_int)]),
    "gpk_predict_batched": (_int, [_vp, _dp, _i64, _dp, _dp, _int]),
    "gpk_lml_batched": (_int, [_vp, _dp, _int, _dp, _dp]),
    "gpk_export": (_int, [_vp, C.POINTER(_i64), C.POINTER(_int), C.POINTER(_int), _dp, _dp, _dp, _dp, _dp]),
    "gpk_import": (_int, [_vp, _dp, _i64, _int, _dp, _dp, _int, _dp, _int, _dbl, _dbl, _dp, _dp]),
    "gpk_model_release": (_int, [_vp]),
    "gpk_split2_rows": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "gpk_split2_rows_f64": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "gpk_split2_rows_f64_absmax": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "gpk_predict_var_inv_split2": (_int, [_vp, _vp, _i64, _int, _dp, _dbl, _vp, _vp, _i64, _vp, _i64, _dbl, _dbl, _vp, _vp]),
    "gpk_predict_mean_var_split2": (_int, [_vp, _vp, _vp, _i64, _int, _int, _dp, _dbl, _dp, _dp, _dp, _vp, _vp, _i64, _vp, _i64,
                                           _dbl, _dbl, _vp, _vp, _dbl, _vp, _vp]),
    "gpk_pack_mean_var": (_int, [_vp, _int, _vp, _vp, _i64, _int, _dp, _vp]),
    "gpk_set_option": (_int, [_vp, C.c_char_p, _int]),
    "gpk_set_option_str": (_int, [_vp, C.c_char_p, C.c_char_p]),
    "gpk_timing": (_int, [_vp, _int]),
    "gpk_kernel_times": (_int, [_vp, _int, _dp, _int, C.POINTER(_int)]),
    "gpk_batch_begin": (_int, [_vp, _int]),
    "gpk_batch_buffer": (_int, [_vp, _vp, _i64]),
    "gpk_batch_end": (_int, [_vp]),
    "gpk_version": (C.c_char_p, []),
    "gpk_gram": (_int, [_vp, _int, _vp, _i64, _int, _dp, _dbl, _dbl, _vp, _i64]),
    "gpk_gram_rows": (_int, [_vp, _int, _vp, _i64, _int, _dp, _dbl, _dbl, _i64, _i64, _vp, _i64]),
    "gpk_cross_gram_t": (_int, [_vp, _int, _vp, _i64, _vp, _i64, _int, _dp, _dbl, _vp, _i64]),
    "gpk_rbf_kernel_grad": (_int, [_vp, _vp, _i64, _vp, _i64, _int, _dp, _dbl, _vp, _vp, _i64]),
    "gpk_potrf": (_int, [_vp, _vp, _i64, _i64, _vp, C.POINTER(_int)]),
    "gpk_leaf_inverses": (_int, [_vp, _vp, _i64, _i64, _vp]),
    "gpk_factor_to_f32": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp]),
    "gpk_potrs": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _i64, _int, _vp]),
    "gpk_potrs_inv": (_int, [_vp, _vp, _i64, _i64, _vp, _i64, _int, _vp]),
    "gpk_wtw": (_int, [_vp, _vp, _i64, _i64, _vp, _i64]),
    "gpk_trsm_lower_left": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _vp, _i64, _i64]),
    "gpk_colsumsq": (_int, [_vp, _int, _vp, _i64, _i64, _i64, _vp]),
    "gpk_predict_mean": (_int, [_vp, _int, _vp, _vp, _i64, _int, _int, _dp, _dbl, _dp, _dp, _vp, _i64, _vp]),
    "gpk_split3": (_int, [_vp, _vp, _i64, _i64, _i64, _vp]),
    "gpk_predict_var_inv_split": (_int, [_vp, _vp, _i64, _int, _dp, _dbl, _vp, _i64, _vp, _i64, _dbl, _dbl, _vp, _vp, _vp]),
    # (host pointers as void*: the serving path passes cached integer addresses)
    "gpk_predict_host": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp, _dbl, _vp, _vp, _vp, _i64, _i64, _dbl, _dbl, _vp,
                                _i64, _vp, _vp]),
    "gpk_predict_host_multi": (_int, [_vp, _int, _vp, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _dbl, _vp,
                                      _i64, _vp, _vp]),
    "gpk_predict_mean_mfma": (_int, [_vp, _vp, _vp, _i64, _int, _int, _dp, _dbl, _dp, _dp, _dp, _vp, _i64, _vp]),
    "gpk_predict_mean_multi": (_int, [_vp, _int, _vp, _vp, _i64, _int, _int, _dp, _dp, _dp, _dp, _vp, _i64, _vp]),
    "gpk_predict_var": (_int, [_vp, _int, _vp, _i64, _int, _dp, _dbl, _vp, _i64, _i64, _vp, _vp, _i64, _dbl,
                               _dbl, _vp, _vp]),
    "gpk_trtri": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp]),
    "gpk_trtri_absmax": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp, _vp]),
    "gpk_tril_to_f32": (_int, [_vp, _vp, _i64, _i64, _vp, _i64]),
    "gpk_predict_var_inv": (_int, [_vp, _int, _vp, _i64, _int, _dp, _dbl, _vp, _i64, _i64, _vp, _i64, _dbl, _dbl,
                                   _vp, _vp]),
    "gpk_lml_terms": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _int, _dp]),
    "gpk_potri": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp]),
    "gpk_lml_grad": (_int, [_vp, _vp, _i64, _int, _dp, _dbl, _dbl, _vp, _int, _vp, _i64, _dp]),
    "gpk_lml_eval": (_int, [_vp, _vp, _i64, _int, _dp, _dbl, _dbl, _dbl, _vp, _int, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _dp, _dp,
                            C.POINTER(C.c_int)]),
    "gpk_gemm_tiles": (_int, [_vp, _int, _int, _int, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _dbl,
                              _dbl, _int]),
}

_lib = None


class GPKError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libgpk error {code}: {message}")
        self.code = code


class NotPositiveDefinite(np.linalg.LinAlgError):
    """Raised where the reference's LAPACK call raises `numpy.linalg.LinAlgError`
    (scipy.linalg.cholesky at sklearn/gaussian_process/_gpr.py:349)."""


def load():
    """Load libgpk.so and declare every prototype.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP runtime (libamdhip64): it must be in the process BEFORE libgpk.so is
    # dlopen'ed, so that libgpk binds to that same runtime (one HIP context, shared streams and allocations).
    # Loading libgpk first would pull in the system runtime and leave torch without a usable device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = os.environ.get("GPK_LIBRARY") or LIB_PATH      # GPK_LIBRARY: an experimental build (see _build.build)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the GP kernels)")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def dptr(a):
    """Host double array -> POINTER(c_double) (the array must stay alive during the call)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)
