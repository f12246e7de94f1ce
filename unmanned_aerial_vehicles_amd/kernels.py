"""Kernel specification objects with the constructor surface the reference uses.

The reference builds its kernels from scikit-learn's `RBF`, `WhiteKernel` and `ConstantKernel`
combined with `+` and `*` (`src/px4/simple_gp.py:167`, `src/px4/gp_trainer.py:163-166`).  These
classes keep that surface — names, arguments, `theta` (log-parameters, fixed ones excluded),
`bounds`, `n_dims`, `clone_with_theta`, the printed form — but hold no arithmetic: evaluation
happens in the HIP kernels.  Supported compositions are exactly the ones the path needs:

    RBF | C*RBF | RBF + White | C*RBF + White          (RBF isotropic or ARD)

`theta` ordering follows scikit-learn (`sklearn/gaussian_process/kernels.py:717-738` for
compound kernels): [log constant_value] + [log length_scale ...] + [log noise_level].
"""
from __future__ import annotations

import copy

import numpy as np


def _bounds_arr(b, n):
    if isinstance(b, str):
        if b != "fixed":
            raise ValueError("bounds must be a pair or 'fixed'")
        return None
    b = np.asarray(b, dtype=np.float64)
    if b.ndim == 1:
        b = np.tile(b.reshape(1, 2), (n, 1))
    return b


class Kernel:
    def __add__(self, other):
        return Sum(self, other)

    def __mul__(self, other):
        return Product(self, other)

    def __radd__(self, other):
        return Sum(ConstantKernel(float(other)), self)

    def __rmul__(self, other):
        return Product(ConstantKernel(float(other)), self)

    @property
    def n_dims(self):
        return self.theta.shape[0]

    def clone_with_theta(self, theta):
        k = copy.deepcopy(self)
        k.theta = theta
        return k

    def _check_bounds_params(self):
        pass

    # components: (sf2, sf2_fixed), (ls, ls_fixed), (noise or None, noise_fixed)
    def components(self):
        return KernelComponents.from_kernel(self)


class ConstantKernel(Kernel):
    def __init__(self, constant_value=1.0, constant_value_bounds=(1e-5, 1e5)):
        self.constant_value = float(constant_value)
        self.constant_value_bounds = constant_value_bounds

    @property
    def fixed(self):
        return isinstance(self.constant_value_bounds, str)

    @property
    def theta(self):
        return np.array([]) if self.fixed else np.log([self.constant_value])

    @theta.setter
    def theta(self, t):
        if not self.fixed:
            self.constant_value = float(np.exp(np.asarray(t).reshape(-1)[0]))

    @property
    def bounds(self):
        b = _bounds_arr(self.constant_value_bounds, 1)
        return np.empty((0, 2)) if b is None else np.log(b)

    def __repr__(self):
        return "{0:.3g}**2".format(np.sqrt(self.constant_value))


class WhiteKernel(Kernel):
    def __init__(self, noise_level=1.0, noise_level_bounds=(1e-5, 1e5)):
        self.noise_level = float(noise_level)
        self.noise_level_bounds = noise_level_bounds

    @property
    def fixed(self):
        return isinstance(self.noise_level_bounds, str)

    @property
    def theta(self):
        return np.array([]) if self.fixed else np.log([self.noise_level])

    @theta.setter
    def theta(self, t):
        if not self.fixed:
            self.noise_level = float(np.exp(np.asarray(t).reshape(-1)[0]))

    @property
    def bounds(self):
        b = _bounds_arr(self.noise_level_bounds, 1)
        return np.empty((0, 2)) if b is None else np.log(b)

    def __repr__(self):
        return "WhiteKernel(noise_level={0:.3g})".format(self.noise_level)


class RBF(Kernel):
    def __init__(self, length_scale=1.0, length_scale_bounds=(1e-5, 1e5)):
        self.length_scale = (np.asarray(length_scale, dtype=np.float64).copy()
                             if np.iterable(length_scale) else float(length_scale))
        self.length_scale_bounds = length_scale_bounds

    @property
    def anisotropic(self):
        return np.iterable(self.length_scale) and len(self.length_scale) > 1

    @property
    def fixed(self):
        return isinstance(self.length_scale_bounds, str)

    @property
    def theta(self):
        if self.fixed:
            return np.array([])
        return np.log(np.atleast_1d(self.length_scale).astype(np.float64))

    @theta.setter
    def theta(self, t):
        if self.fixed:
            return
        v = np.exp(np.asarray(t, dtype=np.float64).reshape(-1))
        self.length_scale = v.copy() if self.anisotropic else float(v[0])

    @property
    def bounds(self):
        n = len(np.atleast_1d(self.length_scale))
        b = _bounds_arr(self.length_scale_bounds, n)
        return np.empty((0, 2)) if b is None else np.log(b)

    def __repr__(self):
        if self.anisotropic:
            return "RBF(length_scale=[{0}])".format(", ".join("{0:.3g}".format(v) for v in self.length_scale))
        return "RBF(length_scale={0:.3g})".format(np.ravel(self.length_scale)[0])


class _Compound(Kernel):
    def __init__(self, k1, k2):
        self.k1, self.k2 = k1, k2

    @property
    def theta(self):
        return np.concatenate([self.k1.theta, self.k2.theta])

    @theta.setter
    def theta(self, t):
        t = np.asarray(t, dtype=np.float64).reshape(-1)
        n1 = self.k1.n_dims
        self.k1.theta = t[:n1]
        self.k2.theta = t[n1:]

    @property
    def bounds(self):
        return np.vstack([self.k1.bounds, self.k2.bounds])


class Sum(_Compound):
    def __repr__(self):
        return "{0} + {1}".format(self.k1, self.k2)


class Product(_Compound):
    def __repr__(self):
        return "{0} * {1}".format(self.k1, self.k2)


C = ConstantKernel


class KernelComponents:
    """Flattened view: k(x, x') = sf2 * exp(-0.5 |(x - x')/ls|^2) + noise * delta(x, x').

    `slots` lists, in theta order, which quantity each free log-parameter drives:
    ("sf2", None) | ("ls", d or None for isotropic) | ("noise", None).
    """

    def __init__(self, sf2, ls, noise, slots, ard):
        self.sf2, self.ls, self.noise, self.slots, self.ard = sf2, ls, noise, slots, ard

    @staticmethod
    def from_kernel(k):
        noise, noise_k = None, None
        body = k
        if isinstance(k, Sum):
            if isinstance(k.k2, WhiteKernel):
                body, noise_k = k.k1, k.k2
            elif isinstance(k.k1, WhiteKernel):
                raise ValueError("put the WhiteKernel last (kernel + WhiteKernel), as the reference does")
            else:
                raise ValueError(f"unsupported kernel sum: {k!r}")
        const_k, rbf_k = None, None
        if isinstance(body, Product):
            if isinstance(body.k1, ConstantKernel) and isinstance(body.k2, RBF):
                const_k, rbf_k = body.k1, body.k2
            else:
                raise ValueError(f"unsupported kernel product: {body!r} (use ConstantKernel * RBF)")
        elif isinstance(body, RBF):
            rbf_k = body
        else:
            raise ValueError(f"unsupported kernel: {k!r}")
        slots = []
        if const_k is not None and not const_k.fixed:
            slots.append(("sf2", None))
        if not rbf_k.fixed:
            if rbf_k.anisotropic:
                slots += [("ls", d) for d in range(len(rbf_k.length_scale))]
            else:
                slots.append(("ls", None))
        if noise_k is not None:
            noise = noise_k.noise_level
            if not noise_k.fixed:
                slots.append(("noise", None))
        sf2 = const_k.constant_value if const_k is not None else 1.0
        return KernelComponents(sf2, np.atleast_1d(np.asarray(rbf_k.length_scale, dtype=np.float64)), noise, slots,
                                rbf_k.anisotropic)

    def ls_vector(self, D):
        if self.ls.size == 1:
            return np.full(D, float(self.ls[0]))
        if self.ls.size != D:
            raise ValueError(f"anisotropic kernel has {self.ls.size} length-scales, X has {D} features")
        return self.ls.copy()

    def map_gradient(self, g, D):
        """g = [g_ls_0..g_ls_{D-1}, g_noise, g_sf2] -> gradient in theta order."""
        out = []
        for kind, d in self.slots:
            if kind == "sf2":
                out.append(g[D + 1])
            elif kind == "noise":
                out.append(g[D])
            elif d is None:
                out.append(float(np.sum(g[:D])))
            else:
                out.append(g[d])
        return np.asarray(out, dtype=np.float64)
