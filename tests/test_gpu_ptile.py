"""One-launch tile Cholesky (`gpk_ptile.hip`: persistent kernel, device-side tile tasks and progress counters) against
the recursive launch chain of the same entry point `gpk_potrf` and against LAPACK on the host, through the C ABI.

Replaces scipy.linalg.cholesky at sklearn/gaussian_process/_gpr.py:349,587.  Bars: factor and tile inverses within 1e-12 /
1e-11 (relative to the largest entry) of the recursion's, L L^T = A to 1e-13 of |A|, W_b L_bb = I to 1e-11.  Every buffer
starts out as NaN (GPK_DEBUG_FILL, conftest.py), so a tile read before it was published fails these comparisons."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from unmanned_aerial_vehicles_amd.device import get_backend
    return get_backend(0)


def _p(t):
    return C.c_void_p(t.data_ptr())


def spd(n, seed, cond_noise=0.05):
    """An RBF Gram matrix + noise (what gpk_potrf sees in production), padded with the identity to whole tiles."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, 6))
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1) if n <= 2048 else None
    if d2 is None:
        G = X @ X.T
        sq = np.diag(G)
        d2 = sq[:, None] + sq[None, :] - 2 * G
    K = np.exp(-0.5 * d2 / 1.5 ** 2)
    K[np.diag_indices(n)] += cond_noise
    npad = (n + 127) // 128 * 128
    A = np.eye(npad)
    A[:n, :n] = K
    return A


def potrf(be, A, ptile):
    import torch
    Np = A.shape[0]
    Ad = be.upload(A)
    winv = be.empty((Np, 128), torch.float64)
    info = C.c_int(0)
    be.check(be.lib.gpk_set_option(be.h, b"ptile", ptile))
    try:
        rc = be.lib.gpk_potrf(be.h, _p(Ad), Np, Np, _p(winv), C.byref(info))
    finally:
        be.check(be.lib.gpk_set_option(be.h, b"ptile", 1))
    return rc, info.value, Ad.cpu().numpy(), winv.cpu().numpy()


@pytest.mark.parametrize("n", [512, 600, 640, 1000, 1920, 2048, 4096])
def test_one_launch_factor_matches_recursion_and_lapack(be, n):
    A = spd(n, n)
    rc1, info1, L1, W1 = potrf(be, A, 1)
    rc0, info0, L0, W0 = potrf(be, A, 0)
    assert rc1 == 0 and rc0 == 0 and info1 == 0 and info0 == 0
    L1, L0 = np.tril(L1), np.tril(L0)
    assert np.isfinite(L1).all() and np.isfinite(W1).all()
    scale = np.abs(L0).max()
    assert np.abs(L1 - L0).max() < 1e-12 * scale
    assert np.abs(W1 - W0).max() < 1e-11 * np.abs(W0).max()
    ref = np.linalg.cholesky(A)
    assert np.abs(L1 - ref).max() < 1e-12 * scale
    assert np.abs(L1 @ L1.T - A).max() < 1e-13 * np.abs(A).max() * 10
    Np = A.shape[0]
    for b in range(Np // 128):
        blk = L1[128 * b:128 * b + 128, 128 * b:128 * b + 128]
        Wb = W1[128 * b:128 * b + 128]
        assert np.abs(Wb @ blk - np.eye(128)).max() < 1e-11
        assert not np.triu(Wb, 1).any()                  # the tile GEMMs that use winv read whole tiles


def test_one_launch_upper_triangle_untouched(be):
    """As the recursion: only the lower triangle of A is written."""
    A = spd(640, 3)
    _, _, L1, _ = potrf(be, A, 1)
    iu = np.triu_indices(640, 1)
    assert np.array_equal(L1[iu], A[iu])


@pytest.mark.parametrize("bad_at", [0, 5, 130, 255, 256, 700, 1023])
def test_one_launch_not_positive_definite(be, bad_at):
    """LAPACK-style 1-based index of the first non-positive pivot, the sweep stays finite (sklearn/_gpr.py:350-358,588-589)."""
    from unmanned_aerial_vehicles_amd import _lib
    A = spd(1024, 7)
    A[bad_at, bad_at] = -1.0
    rc1, info1, L1, W1 = potrf(be, A, 1)
    rc0, info0, _, _ = potrf(be, A, 0)
    assert rc1 == _lib.GPK_NOT_PD and rc0 == _lib.GPK_NOT_PD
    assert info1 == bad_at + 1 == info0
    assert np.isfinite(np.tril(L1)).all() and np.isfinite(W1).all()


@pytest.mark.parametrize("B,n", [(3, 512), (2, 1100), (8, 520)])
def test_one_launch_batched(be, B, n):
    """gpk_batch_begin .. gpk_batch_end: B problems share the launch (config C5's three per-axis GPs), one of them not PD."""
    import torch
    from unmanned_aerial_vehicles_amd import _lib
    As = [spd(n, 100 + b, 0.02 * (b + 1)) for b in range(B)]
    Np = As[0].shape[0]
    if B == 3:
        As[1][200, 200] = -2.0
    Kd = be.upload(np.stack(As))
    winv = be.empty((B, Np, 128), torch.float64)
    info = (C.c_int * 8)()
    be.check(be.lib.gpk_batch_begin(be.h, B))
    try:
        be.check(be.lib.gpk_batch_buffer(be.h, _p(Kd), Np * Np * 8))
        be.check(be.lib.gpk_batch_buffer(be.h, _p(winv), Np * 128 * 8))
        rc = be.lib.gpk_potrf(be.h, _p(Kd), Np, Np, _p(winv), info)
    finally:
        be.lib.gpk_batch_end(be.h)
    assert rc == (_lib.GPK_NOT_PD if B == 3 else _lib.GPK_OK)
    Ls, Ws = Kd.cpu().numpy(), winv.cpu().numpy()
    for b in range(B):
        if B == 3 and b == 1:
            assert info[b] == 201
            continue
        assert info[b] == 0
        ref = np.linalg.cholesky(As[b])
        assert np.abs(np.tril(Ls[b]) - ref).max() < 1e-12 * np.abs(ref).max()
        for t in range(Np // 128):
            blk = ref[128 * t:128 * t + 128, 128 * t:128 * t + 128]
            assert np.abs(Ws[b, 128 * t:128 * t + 128] @ blk - np.eye(128)).max() < 1e-11


def test_one_launch_repeatable_and_large(be):
    """N = 8192 (64 x 64 tiles, more tasks than resident workgroups): bit-identical on a second run, L L^T = A on samples."""
    A = spd(8192, 9)
    _, info, La, Wa = potrf(be, A, 1)
    _, _, Lb, Wb = potrf(be, A, 1)
    assert info == 0
    assert np.array_equal(La, Lb) and np.array_equal(Wa, Wb)
    L = np.tril(La)
    rng = np.random.default_rng(0)
    rows = rng.integers(0, 8192, 40)
    cols = rng.integers(0, 8192, 40)
    for r, c in zip(rows, cols):
        assert abs(L[r] @ L[c] - A[r, c]) < 1e-12
    _, _, L0, W0 = potrf(be, A, 0)
    assert np.abs(L - np.tril(L0)).max() < 1e-12 * np.abs(L).max()
    assert np.abs(Wa - W0).max() < 1e-11 * np.abs(W0).max()


def test_one_launch_above_16384_rows(be):
    """20 480 rows (160 tile columns: two workgroups per CU, whole-tile hand-overs) as ONE launch against the recursion whose halves
    are launches of 10 240 rows (ptile_max_np = 16384, the default until round 5; 24 576 now): the same factor to rounding, compared on the
    device; L L^T = A on sampled entries."""
    import torch
    from unmanned_aerial_vehicles_amd import _lib
    n = 20480
    X = torch.as_tensor(np.random.default_rng(3).standard_normal((n, 9)), device=be.device)
    ls = np.full(9, 2.0)
    K0 = be.empty((n, n), torch.float64)
    be.check(be.lib.gpk_gram(be.h, _lib.GPK_F64, _p(X), n, 9, ls.ctypes.data_as(_lib._dp), 1.0, 0.1001, _p(K0), n))
    winv = be.empty((n, 128), torch.float64)
    info = C.c_int(0)
    out = []
    try:
        for max_np in (24576, 16384):
            be.set_options(ptile_max_np=max_np)
            K = K0.clone()
            assert be.lib.gpk_potrf(be.h, _p(K), n, n, _p(winv), C.byref(info)) == 0 and info.value == 0
            out.append((torch.tril(K), winv.clone()))
    finally:
        be.set_options(ptile_max_np=24576)
    (L1, W1), (L2, W2) = out
    scale = float(L2.abs().max())
    assert float((L1 - L2).abs().max()) < 1e-11 * scale
    assert float((W1 - W2).abs().max()) < 1e-10 * float(W2.abs().max())
    rows = torch.as_tensor(np.random.default_rng(0).integers(0, n, 64), device=be.device)
    cols = torch.as_tensor(np.random.default_rng(1).integers(0, n, 64), device=be.device)
    got = (L1[rows] * L1[cols]).sum(1)
    want = K0[torch.maximum(rows, cols), torch.minimum(rows, cols)]
    assert float((got - want).abs().max()) < 1e-11


@pytest.mark.parametrize("n", [640, 1000, 2048, 3000])
def test_one_launch_handovers_on_and_off(be, n):
    """The 16-column hand-overs (the two tiles under a diagonal tile follow its factorisation block row by block row -
    forward substitution with L_jj's rows and the W_bb - and publish their own 16-column blocks) against the same launch
    with whole-tile hand-overs (ptile_prog_max_nt = 0: the tile under the diagonal multiplies by the finished W_jj):
    different arithmetic for those tiles, the same factor to rounding; both against LAPACK."""
    A = spd(n, 31 + n)
    be.check(be.lib.gpk_set_option(be.h, b"ptile_prog_max_nt", 0))
    try:
        rc0, info0, L0, W0 = potrf(be, A, 1)
    finally:
        be.check(be.lib.gpk_set_option(be.h, b"ptile_prog_max_nt", 128))
    rc1, info1, L1, W1 = potrf(be, A, 1)
    assert rc0 == 0 and rc1 == 0 and info0 == 0 and info1 == 0
    L0, L1 = np.tril(L0), np.tril(L1)
    ref = np.linalg.cholesky(A)
    scale = np.abs(ref).max()
    assert np.abs(L1 - ref).max() < 1e-12 * scale and np.abs(L0 - ref).max() < 1e-12 * scale
    assert np.abs(L1 - L0).max() < 1e-12 * scale
    assert np.abs(W1 - W0).max() < 1e-11 * np.abs(W0).max()
    # any number of followers per column (default 8): a follower's tile goes through the forward substitution instead of the
    # product with the finished inverse - the same factor to rounding whatever the count
    for rows in (1, 2, 3, 5):
        be.check(be.lib.gpk_set_option(be.h, b"ptile_prog_rows", rows))
        try:
            _, info2, L2, W2 = potrf(be, A, 1)
        finally:
            be.check(be.lib.gpk_set_option(be.h, b"ptile_prog_rows", 8))
        assert info2 == 0 and np.abs(np.tril(L2) - ref).max() < 1e-12 * scale, rows
        assert np.abs(W2 - W0).max() < 1e-11 * np.abs(W0).max(), rows
    # two resident workgroups per CU instead of one (the launch's form above 96 tile columns): who runs a task changes, what it
    # computes does not - the factor and the tile inverses are bit-identical
    be.check(be.lib.gpk_set_option(be.h, b"ptile_single_max_nt", 0))
    try:
        _, info3, L3, W3 = potrf(be, A, 1)
    finally:
        be.check(be.lib.gpk_set_option(be.h, b"ptile_single_max_nt", 96))
    assert info3 == 0 and np.array_equal(np.tril(L3), L1) and np.array_equal(W3, W1)
    # the 128-register build of the launch (the small launches run a 256-register build with two k-tiles in flight in the
    # off-diagonal k-loops): the same arithmetic in the same order - bit-identical
    be.set_options(ptile_sr=0)
    try:
        _, info4, L4, W4 = potrf(be, A, 1)
    finally:
        be.set_options(ptile_sr=1)
    assert info4 == 0 and np.array_equal(np.tril(L4), L1) and np.array_equal(W4, W1)


@pytest.mark.parametrize("n", [640, 2048, 3000])
def test_xcd_aware_dealing_gives_the_same_bits(be, n):
    """The optional XCD-aware dealing (`ptile_xcd`: one task queue per XCD - tile rows round-robin, or R x C groups per queue -
    instead of ONE global ticket) changes WHO runs a task and in which order tasks are taken, never what a task computes: the
    factor and the tile inverses are bit-identical, with one and with two workgroups per CU, and a not-PD matrix is still
    reported (the queues must drain when the launch gives up on a pivot)."""
    A = spd(n, n + 7)
    _, info0, L0, W0 = potrf(be, A, 1)
    assert info0 == 0
    opts = dict(ptile_xcd=0, ptile_xcd_min_nt=56, ptile_grp_rows=8, ptile_grp_cols=4, ptile_single_max_nt=96)
    try:
        for mode, rows, cols, single in ((1, 8, 4, 96), (2, 8, 4, 96), (2, 4, 8, 0), (1, 8, 4, 0), (2, 2, 3, 96)):
            be.set_options(ptile_xcd=mode, ptile_xcd_min_nt=0, ptile_grp_rows=rows, ptile_grp_cols=cols, ptile_single_max_nt=single)
            _, info, L, W = potrf(be, A, 1)
            assert info == 0 and np.array_equal(np.tril(L), np.tril(L0)) and np.array_equal(W, W0), (mode, rows, cols, single)
        B = A.copy()
        B[300, 300] = -1.0
        be.set_options(ptile_xcd=2, ptile_grp_rows=8, ptile_grp_cols=4, ptile_single_max_nt=96)
        rc, info, _, _ = potrf(be, B, 1)
        assert rc == 1 and info == 301
    finally:
        be.set_options(**opts)
