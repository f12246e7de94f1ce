"""Pins the CPU oracle (oracle/gp_oracle.py) against the golden fixtures produced by
scikit-learn 1.7.2 and by the reference's own modules (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from conftest import relerr
from oracle import gp_oracle as O

TOL = 1e-10  # oracle vs sklearn at fixed theta (same LAPACK calls, fp64)


def _fit(csv_data, D, cols, ls, noise, jitter=1e-4, normalize_y=True, sf2=1.0):
    X = csv_data["X10"][:, :D]
    Y = csv_data["Y6"][:, cols]
    return O.fit_fixed(X, Y, ls, sf2, noise, jitter, normalize_y)


@pytest.mark.parametrize("name,D,cols,ls,noise", [
    ("ka1", 10, slice(0, 6), 0.5, 0.1),
    ("ka2", 9, slice(3, 6), 0.5, 0.1),
    ("ka2b", 9, slice(3, 6), 0.114, 0.35),
])
def test_fixed_theta_models(csv_data, ka, name, D, cols, ls, noise):
    st = _fit(csv_data, D, cols, ls, noise)
    assert relerr(st.alpha, ka[f"{name}_alpha"]) < 1e-9
    assert relerr(np.diag(st.L), ka[f"{name}_Ldiag"]) < TOL
    assert relerr(np.linalg.norm(st.L), ka[f"{name}_Lfro"]) < TOL
    assert relerr(st.L[ka[f"{name}_Lrows_idx"]], ka[f"{name}_Lrows"]) < TOL
    assert relerr(st.y_mean, ka[f"{name}_ymean"]) < 1e-14
    assert relerr(st.y_std, ka[f"{name}_ystd"]) < 1e-14
    assert abs(O.log_marginal_likelihood(st) - ka[f"{name}_lml"]) < 1e-9 * abs(ka[f"{name}_lml"])
    mean, std = O.predict(st, csv_data["Xq10"][:, :D], return_std=True)
    assert relerr(mean, ka[f"{name}_mean"]) < 1e-10
    assert relerr(std, ka[f"{name}_std"]) < 1e-9


def test_d9_equals_d10(csv_data):
    """yaw_rate is numerically zero in the north-star CSV: dropping it is bit-identical."""
    a = O.fit_fixed(csv_data["X10"], csv_data["Y6"], 0.5, 1.0, 0.1, 1e-4)
    b = O.fit_fixed(csv_data["X10"][:, :9], csv_data["Y6"], 0.5, 1.0, 0.1, 1e-4)
    assert np.array_equal(a.alpha, b.alpha)


def test_gram_rows(csv_data, ka):
    K = O.rbf_gram(csv_data["X10"][:, :9], 0.5, 1.0, diag_add=0.1)
    assert relerr(K[ka["ka2_Krows_idx"]], ka["ka2_Krows"]) < 1e-14


@pytest.mark.parametrize("name,ls,noise", [("ka2", 0.5, 0.1), ("ka2b", 0.114, 0.35)])
def test_lml_gradient_iso(csv_data, ka, name, ls, noise):
    st = _fit(csv_data, 9, slice(3, 6), ls, noise)
    g = O.lml_gradient(st, ard=False)
    assert relerr(g, ka[f"{name}_grad"]) < 1e-9


def test_ard_model(csv_data, ka):
    X = csv_data["X10"][:, :9]
    y = csv_data["Y6"][:, 3]
    st = O.fit_fixed(X, y, np.ones(9), 1.0, 0.01, 1e-6, normalize_y=False)
    assert abs(O.log_marginal_likelihood(st) - ka["ka6_lml"]) < 1e-9 * abs(ka["ka6_lml"])
    assert relerr(O.lml_gradient(st, ard=True), ka["ka6_grad"]) < 1e-8
    mean, std = O.predict(st, csv_data["Xq10"][:, :9], return_std=True)
    assert relerr(mean[:, 0], ka["ka6_mean"]) < 1e-9
    assert relerr(std[:, 0], ka["ka6_std"]) < 1e-8
    st = O.fit_fixed(X, y, ka["ka6b_ls"], 1.0, 0.05, 1e-6, normalize_y=False)
    assert abs(O.log_marginal_likelihood(st) - ka["ka6b_lml"]) < 1e-9 * abs(ka["ka6b_lml"])
    assert relerr(O.lml_gradient(st, ard=True), ka["ka6b_grad"]) < 1e-8
    assert relerr(st.alpha[:, 0], ka["ka6b_alpha"]) < 1e-9
    mean, std = O.predict(st, csv_data["Xq10"][:, :9], return_std=True)
    assert relerr(mean[:, 0], ka["ka6b_mean"]) < 1e-9
    assert relerr(std[:, 0], ka["ka6b_std"]) < 1e-8


def test_reference_trained_model(csv_data, ka):
    """KA3: the reference's SimpleQuadrotorGP.train_gp() result, re-evaluated by the oracle at
    the reference's final theta (the optimiser path itself is not bit-stable)."""
    ls, noise = np.exp(ka["ka3_theta"])
    st = O.fit_fixed(csv_data["X10"], csv_data["Y6"], ls, 1.0, noise, 1e-4)
    assert abs(O.log_marginal_likelihood(st) - ka["ka3_lml"]) < 1e-9 * abs(ka["ka3_lml"])
    assert relerr(st.y_mean, ka["ka3_ymean"]) < 1e-14
    m, v = O.predict_residual(st, csv_data["X10"][24, :6], csv_data["X10"][24, 6:])
    assert relerr(m, ka["ka3_pred_mean"]) < 1e-9
    assert relerr(v, ka["ka3_pred_var"]) < 1e-8
    mean, std = O.predict(st, csv_data["Xq10"], return_std=True)
    assert relerr(mean, ka["ka3_mean"]) < 1e-9
    assert relerr(std, ka["ka3_std"]) < 1e-8
    D = O.build_gp_residuals(st, ka["ka3_hor_X"], ka["ka3_hor_U"], float(ka["ka3_hor_dt"]))
    assert relerr(D, ka["ka3_hor_D"]) < 1e-9
    assert abs(np.mean(np.sqrt(v)) - ka["ka3_uncertainty"]) < 1e-10


def test_package_gp(csv_data, ka):
    g = O.PackageGPOracle(1.0, 1.0, 0.01).fit(csv_data["X10"][:, :9], csv_data["Y6"][:, 3:6])
    assert abs(g.log_marginal_likelihood() - ka["ka5_lml"]) < 1e-10 * abs(ka["ka5_lml"])
    assert relerr(g.alpha, ka["ka5_alpha"]) < 1e-10
    assert relerr(np.diag(g.L), ka["ka5_Ldiag"]) < 1e-12
    mean, var = g.predict(csv_data["Xq10"][:, :9])
    assert relerr(mean, ka["ka5_mean"]) < 1e-10
    assert relerr(var, ka["ka5_var"]) < 1e-9


def test_c2_synthetic(ka):
    X, Y, Xq = O.synthetic_problem(4096, 1024)
    st = O.fit_fixed(X, Y, 2.0, 1.0, 0.1, 1e-4)
    assert relerr(np.diag(st.L), ka["c2_Ldiag"]) < 1e-10
    assert relerr(st.alpha, ka["c2_alpha"]) < 1e-8
    assert abs(O.log_marginal_likelihood(st) - ka["c2_lml"]) < 1e-9 * abs(ka["c2_lml"])
    mean, std = O.predict(st, Xq, return_std=True)
    assert relerr(mean, ka["c2_mean"]) < 1e-9
    assert relerr(std, ka["c2_std"]) < 1e-8
