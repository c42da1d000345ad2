"""Native operator objects: callback providers the engine recognises.

The reference's solver takes four opaque callables (``f, g, jac_f,
prox_wsum_g``; zfista/proximal_gradient.py:311-316) and its problem library
supplies them as bound methods (zfista/problems.py:140-150).  The objects here
keep that contract - ``prob.f(x)`` etc. are ordinary callables on NumPy arrays,
evaluated on the GPU - and additionally carry a descriptor.  When
``zfista_amd.minimize_proximal_gradient`` is handed the four bound methods of
ONE such object it runs the device-resident fused path instead of calling them.

Problem data lives in HBM (torch CUDA tensors are accepted as-is; NumPy arrays
are uploaded once).  There is no host implementation of any of these methods.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _to_device(a, name):
    """float64 contiguous CUDA tensor from a NumPy array or a torch tensor."""
    import torch

    _lib.require_gpu()
    if isinstance(a, torch.Tensor):
        t = a
        if not t.is_cuda:
            t = t.cuda()
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).cuda()
    if t.dtype != torch.float64:
        raise TypeError(f"{name} must be float64 (the reference path is float64 throughout)")
    return t.contiguous()


def _as_host(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


class NativeProblem:
    """Marker base: a single-objective problem with a device descriptor."""

    n_features: int

    def callbacks(self):
        return self.f, self.g, self.jac_f, self.prox_wsum_g

    def minimize_proximal_gradient(self, x0, **kwargs):
        """Bound-method form, as zfista/problems.py:140-150."""
        from .proximal_gradient import minimize_proximal_gradient

        return minimize_proximal_gradient(self.f, self.g, self.jac_f, self.prox_wsum_g, x0, **kwargs)

    # -- shared g / prox: lam * |x|_1 (+ box) ------------------------------------------
    def g(self, x):
        x = _as_host(x)
        if self._has_box() and ((x < self.box[0]).any() or (x > self.box[1]).any()):
            return np.float64(np.inf)   # zfista/problems.py:104-106
        return self._eval_fg(x)[1]

    def prox_wsum_g(self, weight, x):
        x = _as_host(x)
        out = np.empty_like(x)
        lib = _lib.require_gpu()
        _lib.check(lib.zf_host_prox_l1_box(C.c_void_p(_lib.ptr(out)), C.c_void_p(_lib.ptr(x)),
                                           float(self.lam * weight), self.box[0], self.box[1], x.size),
                   "zf_host_prox_l1_box")
        return out

    def _has_box(self):
        return not (self.box[0] == -np.inf and self.box[1] == np.inf)


class DiagQuadL1(NativeProblem):
    r"""f(x) = 1/2 \sum_i d_i (x_i - c_i)^2,  g(x) = lam \|x\|_1 (+ optional box).

    The single-objective analogue of the diagonal-gradient problems of
    zfista/problems.py:193-205 (BASELINE cfg2 / the headline metric).  With
    ``group`` set, ``d`` and ``c`` are this rank's contiguous shard of a
    decision vector partitioned across the ranks of that process group.
    """

    kind = _lib.ZF_PROBLEM_DIAG_QUAD_L1
    separable = True   # f is a sum over elements: chains of trials per pass, and acceptance="resolved"

    def __init__(self, d, c, lam, bounds=None, group=None):
        self.d = _to_device(d, "d")
        self.c = _to_device(c, "c")
        if self.d.ndim != 1 or self.d.shape != self.c.shape:
            raise ValueError("d and c must be 1-D of equal length")
        self.lam = float(lam)
        self.box = (-np.inf, np.inf) if bounds is None else (float(bounds[0]), float(bounds[1]))
        self.n_features = int(self.d.numel())
        self.group = group

    def _eval_fg(self, x):
        import torch

        if x.size != self.n_features:
            raise ValueError(f"len(x) should be equal to n_features, got {x}.")
        lib = _lib.require_gpu()
        xd = torch.from_numpy(x).cuda()
        out = np.zeros(2)
        _lib.check(lib.zf_eval_diag_l1(C.c_void_p(xd.data_ptr()), C.c_void_p(self.d.data_ptr()),
                                       C.c_void_p(self.c.data_ptr()), self.lam, x.size,
                                       C.c_void_p(_lib.ptr(out)),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "zf_eval_diag_l1")
        return np.float64(out[0]), np.float64(out[1])

    def f(self, x):
        return self._eval_fg(_as_host(x))[0]

    def jac_f(self, x):
        x = _as_host(x)
        if x.size != self.n_features:
            raise ValueError(f"len(x) should be equal to n_features, got {x}.")
        out = np.empty_like(x)
        lib = _lib.require_gpu()
        _lib.check(lib.zf_host_diag_grad(C.c_void_p(_lib.ptr(out)), C.c_void_p(_lib.ptr(x)),
                                         C.c_void_p(self.d.data_ptr()), C.c_void_p(self.c.data_ptr()),
                                         x.size), "zf_host_diag_grad")
        return out

    def _descriptor(self):
        from .comm import rank_world

        rank, world = rank_world(self.group)
        fields = dict(kind=self.kind, world=world, rank=rank, n=self.n_features, m_rows=0,
                      d=self.d.data_ptr(), c=self.c.data_ptr(), A=None, b=None,
                      scale=0.5, lam=self.lam, box_lo=self.box[0], box_hi=self.box[1])
        return fields, (self.d, self.c)


class LeastSquaresL1(NativeProblem):
    r"""f(x) = scale \|Ax - b\|^2,  g(x) = lam \|x\|_1 (+ optional box); A dense row-major.

    The LASSO closures of tests/test_proximal_gradient.py:49-61,81-97 are the
    ``scale = 1/6`` member; BASELINE cfg1 / cfg3 use ``scale = 1/2``.
    """

    kind = _lib.ZF_PROBLEM_LEAST_SQUARES_L1

    def __init__(self, A, b, lam, scale=0.5, bounds=None, group=None, shard="columns"):
        """With ``group`` set and ``shard="columns"`` (default), ``A`` is this rank's column block A_p
        (m x n_p, row-major) of a matrix whose columns - and the decision vector - are partitioned
        over the ranks of that process group; ``b`` is replicated; the solve exchanges the m-vector
        A_p x_p once per trial.  ``shard="rows"``: ``A`` (m_p x n) and ``b`` (m_p) are this rank's ROW
        block, x is replicated on every rank (pass the whole x0) and the n-vector A_p^T r_p is
        exchanged instead - the layout for tall matrices.  ``f`` / ``jac_f`` as plain callables
        refer to the local block only."""
        if shard not in ("columns", "rows"):
            raise ValueError("shard must be 'columns' or 'rows'")
        self.shard = shard
        self.A = _to_device(A, "A")
        self.b = _to_device(b, "b")
        if self.A.ndim != 2 or self.b.ndim != 1 or self.A.shape[0] != self.b.shape[0]:
            raise ValueError("A must be (m, n) and b (m,)")
        self.lam, self.scale = float(lam), float(scale)
        self.box = (-np.inf, np.inf) if bounds is None else (float(bounds[0]), float(bounds[1]))
        self.m_rows, self.n_features = int(self.A.shape[0]), int(self.A.shape[1])
        self.group = group

    def _ls(self, x, want_grad):
        x = _as_host(x)
        if x.size != self.n_features:
            raise ValueError(f"len(x) should be equal to n_features, got {x}.")
        lib = _lib.require_gpu()
        fval = C.c_double(0.0)
        grad = np.empty_like(x) if want_grad else None
        _lib.check(lib.zf_ls_eval(C.c_void_p(self.A.data_ptr()), C.c_void_p(self.b.data_ptr()),
                                  self.m_rows, self.n_features, self.scale, C.c_void_p(_lib.ptr(x)),
                                  C.byref(fval), C.c_void_p(_lib.ptr(grad)) if want_grad else None),
                   "zf_ls_eval")
        return np.float64(fval.value), grad

    def f(self, x):
        return self._ls(x, False)[0]

    def jac_f(self, x):
        return self._ls(x, True)[1]

    def _eval_fg(self, x):
        lib = _lib.require_gpu()
        s = C.c_double(0.0)
        _lib.check(lib.zf_host_asum(C.c_void_p(_lib.ptr(x)), x.size, C.byref(s)), "zf_host_asum")
        return None, np.float64(self.lam * s.value)

    def _descriptor(self):
        from .comm import rank_world

        rank, world = rank_world(self.group)
        fields = dict(kind=self.kind, world=world, rank=rank, n=self.n_features, m_rows=self.m_rows,
                      row_sharded=int(self.shard == "rows" and world > 1),
                      d=None, c=None, A=self.A.data_ptr(), b=self.b.data_ptr(),
                      scale=self.scale, lam=self.lam, box_lo=self.box[0], box_hi=self.box[1])
        return fields, (self.A, self.b)


class BlurHaarL1(NativeProblem):
    r"""Operator-form LASSO: f(x) = scale \|B W^{-1} x - b\|^2,  g(x) = lam \|x\|_1 (+ optional box), with B the
    correlation with ``kernel`` (odd size <= 15, symmetric boundary - ``scipy.signal.correlate2d(..., mode="same",
    boundary="symm")``) and W one orthonormal Haar level (``pywt.dwt2(.., "haar")``, coefficients flattened as
    [cA, cH, cV, cD]): the image-deblurring problem of the reference's ``examples/cameraman.ipynb`` (cells 6-11,
    ``l1_ratio`` = lam, scale = 1).  ``jac_f`` applies B itself as its adjoint, as the notebook does.

    The callbacks keep the notebook's types - ``f`` and ``g`` return arrays of ONE value, ``jac_f`` a (1, n) array
    (m = 1 by zfista/proximal_gradient.py:143) - and so do ``fun`` / ``allfuns`` of a solve.  Handed to
    ``minimize_proximal_gradient`` as its four bound methods, the solve runs device-resident: the Haar levels are
    folded into the tile loads / epilogues of the two correlation kernels (csrc/zf_kernels_op.h), B W^-1 y comes by
    linearity from the cached B W^-1 x_k, B W^-1 x_{k-1}; no host synchronisation per iteration."""

    kind = _lib.ZF_PROBLEM_BLUR_HAAR_L1
    array_valued = True      # f, g -> (1,) arrays; jac_f -> (1, n)

    def __init__(self, kernel, observed, l1_ratio, scale=1.0, bounds=None):
        self.taps = _to_device(kernel, "kernel")
        self.b = _to_device(observed, "observed")
        if self.taps.ndim != 2 or self.taps.shape[0] != self.taps.shape[1] or self.taps.shape[0] % 2 != 1 or self.taps.shape[0] > 15:
            raise ValueError("kernel must be square with an odd size of at most 15")
        if self.b.ndim != 2 or self.b.shape[0] % 2 or self.b.shape[1] % 2:
            raise ValueError("the observed image must be 2-D with even sides")
        if self.taps.shape[0] // 2 >= min(self.b.shape):
            raise ValueError("the kernel is too large for the image")
        self.k = int(self.taps.shape[0])
        self.h, self.w = int(self.b.shape[0]), int(self.b.shape[1])
        self.lam, self.scale = float(l1_ratio), float(scale)
        self.box = (-np.inf, np.inf) if bounds is None else (float(bounds[0]), float(bounds[1]))
        self.n_features = self.h * self.w
        self.group = None

    def _op(self, x, want_grad):
        x = _as_host(x).reshape(-1)
        if x.size != self.n_features:
            raise ValueError(f"len(x) should be equal to n_features, got {x}.")
        lib = _lib.require_gpu()
        fval = C.c_double(0.0)
        grad = np.empty_like(x) if want_grad else None
        _lib.check(lib.zf_op_eval(C.c_void_p(self.taps.data_ptr()), self.k, C.c_void_p(self.b.data_ptr()), self.h, self.w,
                                  self.scale, C.c_void_p(_lib.ptr(x)), C.byref(fval),
                                  C.c_void_p(_lib.ptr(grad)) if want_grad else None), "zf_op_eval")
        return np.float64(fval.value), grad

    def f(self, x):
        return np.array([self._op(x, False)[0]])

    def jac_f(self, x):
        return self._op(x, True)[1].reshape(1, -1)

    def g(self, x):
        return np.array([NativeProblem.g(self, np.asarray(x).reshape(-1))])

    def prox_wsum_g(self, weight, x):
        # (weight is lr - a float - for m = 1, :148; tolerate the 1-element array a caller may pass)
        return NativeProblem.prox_wsum_g(self, float(np.asarray(weight).reshape(-1)[0]), np.asarray(x).reshape(-1))

    def _eval_fg(self, x):
        lib = _lib.require_gpu()
        s = C.c_double(0.0)
        _lib.check(lib.zf_host_asum(C.c_void_p(_lib.ptr(x)), x.size, C.byref(s)), "zf_host_asum")
        return None, np.float64(self.lam * s.value)

    def _descriptor(self):
        fields = dict(kind=self.kind, world=1, rank=0, n=self.n_features, m_rows=self.n_features, row_sharded=0,
                      d=None, c=None, A=None, b=self.b.data_ptr(), scale=self.scale, lam=self.lam,
                      box_lo=self.box[0], box_hi=self.box[1], op_h=self.h, op_w=self.w, op_taps=self.taps.data_ptr(),
                      op_k=self.k)
        return fields, (self.taps, self.b)


def match_native(f, g, jac_f, prox_wsum_g):
    """The NativeProblem whose four bound methods these are, else None."""
    owner = getattr(f, "__self__", None)
    if not isinstance(owner, NativeProblem):
        return None
    want = ("f", "g", "jac_f", "prox_wsum_g")
    for cb, name in zip((f, g, jac_f, prox_wsum_g), want):
        if getattr(cb, "__self__", None) is not owner:
            return None
        if getattr(cb, "__func__", None) is not getattr(type(owner), name):
            return None
    return owner


# ---------------------------------------------------------------------------
# multi-objective problem family of zfista/problems.py (shifted l1 + box)
# ---------------------------------------------------------------------------
class Problem:
    """Mirror of zfista.problems.Problem (problems.py:25-150): F_i = f_i + g_i with
    g_i(x) = l1_ratios[i] * |x - l1_shifts[i]|_1 plus an optional box.

    ``g`` and ``prox_wsum_g`` are NumPy-callable and evaluated on the GPU
    (``zf_mo_eval_F`` / ``zf_mo_prox_host``).  Subclasses with a built-in device
    ``f`` / ``jac_f`` (``JOS1``, ``FDS``) are run by ``minimize_proximal_gradient``
    with everything but SciPy's dual search on the GPU.
    """

    _kind = _lib.ZF_MO_GENERIC

    def __init__(self, n_features, n_objectives, l1_ratios=None, l1_shifts=None, bounds=None, group=None):
        # group (torch.distributed): the decision vector is split over its ranks in contiguous
        # blocks; n_features stays the length of the WHOLE vector, x0 / res.x are this rank's block
        self.group = group
        self.n_features = int(n_features)
        self.n_objectives = int(n_objectives)
        self.l1_ratios = None if l1_ratios is None else np.array(l1_ratios, dtype=np.float64)
        self.l1_shifts = np.zeros(n_objectives) if l1_shifts is None else np.array(l1_shifts, dtype=np.float64)
        self.bounds = bounds   # scalars or per-coordinate arrays of n_features (problems.py:69-70)
        self.name = self._generate_name()
        self._eng = None

    def _generate_name(self):   # problems.py:81-91
        parts = [type(self).__name__, f"n_{self.n_features}"]
        if self.l1_ratios is not None:
            parts.append("l1_ratios_" + "_".join(map(str, self.l1_ratios)))
            parts.append("l1_shifts_" + "_".join(map(str, self.l1_shifts)))
        if self.bounds is not None:
            parts.append("bounds_" + "_".join(map(str, [self.bounds[0], self.bounds[1]])))
        return "_".join(parts)

    def _engine(self):
        from .multiobjective import MoEngine

        if self._eng is None or self._eng.h is None:
            lo, hi = self.shard_bounds()
            arrays = self.bounds is not None and (np.ndim(self.bounds[0]) > 0 or np.ndim(self.bounds[1]) > 0)
            self._eng = MoEngine(self._kind, self.n_objectives, hi - lo, self.l1_ratios,
                                 self.l1_shifts, None if arrays else self.bounds, group=self.group,
                                 n_global=self.n_features, offset=lo)
            if arrays:   # per-coordinate box: this rank's block of each bound vector
                full = [np.broadcast_to(np.asarray(b, dtype=np.float64), (self.n_features,)) for b in self.bounds]
                self._eng.set_bounds(full[0][lo:hi], full[1][lo:hi])
        return self._eng

    def shard_bounds(self):
        """[lo, hi): this rank's block of the decision vector (the whole vector without a group)."""
        if self.group is None:
            return 0, self.n_features
        from .comm import rank_world

        rank, world = rank_world(self.group)
        return rank * self.n_features // world, (rank + 1) * self.n_features // world

    def _check_len(self, x):
        lo, hi = self.shard_bounds()
        if hi - lo != len(x):
            raise ValueError(f"len(x) should be equal to n_features, got {x}.")   # problems.py:102-103

    def g(self, x):
        self._check_len(x)
        eng = self._engine()
        eng.put(2, x)
        return eng.eval_F(2, builtin_f=False)[1]

    def prox_wsum_g(self, weight, x):
        self._check_len(x)
        return self._engine().prox_host(weight, x)

    def f(self, x):
        raise NotImplementedError

    def jac_f(self, x):
        raise NotImplementedError

    def callbacks(self):
        return self.f, self.g, self.jac_f, self.prox_wsum_g

    def minimize_proximal_gradient(self, x0, **kwargs):   # problems.py:140-150
        from .proximal_gradient import minimize_proximal_gradient

        return minimize_proximal_gradient(self.f, self.g, self.jac_f, self.prox_wsum_g, x0, **kwargs)


class _BuiltinProblem(Problem):
    _host_f = False

    def f(self, x):
        self._check_len(x)
        eng = self._engine()
        eng.put(2, x)
        return eng.eval_F(2)[0]

    def jac_f(self, x):
        self._check_len(x)
        eng = self._engine()
        saved = eng.get(1)
        eng.put(1, x)
        eng.prepare()
        J = eng.get_jac()
        eng.put(1, saved)
        return J


class JOS1(_BuiltinProblem):
    """f_1 = |x|^2 / n, f_2 = |x - 2|^2 / n   (zfista/problems.py:153-205)."""

    _kind = _lib.ZF_MO_JOS1

    def __init__(self, n_features=5, l1_ratios=None, l1_shifts=None, bounds=None, group=None):
        super().__init__(n_features, 2, l1_ratios, l1_shifts, bounds, group=group)


class FDS(_BuiltinProblem):
    """Fliege-Drummond-Svaiter test problem, m = 3   (zfista/problems.py:267-328)."""

    _kind = _lib.ZF_MO_FDS

    def __init__(self, n_features=10, l1_ratios=None, l1_shifts=None, bounds=None, group=None):
        super().__init__(n_features, 3, l1_ratios, l1_shifts, bounds, group=group)


# The remaining families of zfista/problems.py have n_features between 3 and 30: their f / jac_f
# are a handful of scalar operations, evaluated on the host (a kernel launch would cost more than
# the arithmetic); g, prox_wsum_g and the solver's own vector work run on the GPU like for every
# Problem.  Each follows the reference's CODE where it differs from its docstring.
class _HostProblem(Problem):
    # f / jac_f are host NumPy; g, prox and every O(n) expression of the dual run in the device
    # engine (multiobjective.solve_native), one host round trip per dual evaluation
    _host_f = True

    def _x(self, x):
        self._check_len(x)
        return np.asarray(x, dtype=np.float64)


class SD(_HostProblem):
    """Stadler-Dauer, n = 4, m = 2, box (1e-6, inf)   (zfista/problems.py:208-264)."""

    def __init__(self):
        super().__init__(4, 2, bounds=(1e-6, np.inf))

    def f(self, x):
        x = self._x(x)
        r2 = np.sqrt(2)
        return np.array([2 * x[0] + r2 * x[1] + r2 * x[2] + x[3],
                         2 / x[0] + 2 * r2 / x[1] + 2 * r2 / x[2] + 2 / x[3]])   # :251 (2 / x_4 as coded)

    def jac_f(self, x):
        x = np.asarray(x, dtype=np.float64)
        r2 = np.sqrt(2)
        return np.vstack((np.array([2, r2, r2, 1]),
                          np.array([-2 / x[0] ** 2, -2 * r2 / x[1] ** 2, -2 * r2 / x[2] ** 2, -2 / x[3] ** 2])))


class ZDT1(_HostProblem):
    """Zitzler-Deb-Thiele 1, m = 2, box (1e-6, inf)   (zfista/problems.py:331-386)."""

    def __init__(self, n_features=30):
        super().__init__(n_features, 2, bounds=(1e-6, np.inf))

    def f(self, x):
        x = self._x(x)
        h = 1 + 9 / (self.n_features - 1) * np.sum(x[1:])
        return np.array([x[0], h * (1 - np.sqrt(x[0] / h))])

    def jac_f(self, x):
        x = self._x(x)
        n = self.n_features
        h = 1 + 9 / (n - 1) * np.sum(x[1:])
        j1 = np.zeros(n)
        j1[0] = 1
        j2 = np.full(n, 9 * (2 - np.sqrt(x[0] / h)) / 2 / (n - 1))   # :381-383 as coded (the derivative)
        j2[0] = -np.sqrt(h / x[0]) / 2
        return np.vstack((j1, j2))


class TOI4(_HostProblem):
    """Toint 4, n = 4, m = 2   (zfista/problems.py:389-448)."""

    def __init__(self, l1_ratios=None, l1_shifts=None, bounds=None):
        super().__init__(4, 2, l1_ratios, l1_shifts, bounds)

    def f(self, x):
        x = self._x(x)
        return np.array([x[0] ** 2 + x[1] ** 2 + 1, 0.5 * ((x[0] - x[1]) ** 2 + (x[2] - x[3]) ** 2) + 1])

    def jac_f(self, x):
        x = self._x(x)
        a, b = x[0] - x[1], x[2] - x[3]
        return np.array([[2 * x[0], 2 * x[1], 0.0, 0.0], [a, -a, b, -b]])


class TRIDIA(_HostProblem):
    """Toint tridiagonal, n = 3, m = 3   (zfista/problems.py:451-514)."""

    def __init__(self, l1_ratios=None, l1_shifts=None, bounds=None):
        super().__init__(3, 3, l1_ratios, l1_shifts, bounds)

    def f(self, x):
        x = self._x(x)
        return np.array([(2 * x[0] - 1) ** 2, 2 * (2 * x[0] - x[1]) ** 2, 3 * (2 * x[1] - x[2]) ** 2])

    def jac_f(self, x):
        x = self._x(x)
        return np.array([[8 * x[0] - 4, 0, 0],
                         [16 * x[0] - 8 * x[1], 4 * x[1] - 8 * x[0], 0],
                         [0, 24 * x[1] - 12 * x[2], 6 * x[2] - 12 * x[1]]], dtype=np.float64)


class LinearFunctionRank1(_HostProblem):
    """f_i = (i <(1..n), x> - 1)^2, i = 1..m   (zfista/problems.py:517-575)."""

    def __init__(self, n_features=10, n_objectives=4, l1_ratios=None, l1_shifts=None, bounds=None):
        super().__init__(n_features, n_objectives, l1_ratios, l1_shifts, bounds)
        self.range_n_objectives = np.arange(1, self.n_objectives + 1)
        self.range_n_features = np.arange(1, self.n_features + 1)

    def f(self, x):
        x = self._x(x)
        return (self.range_n_objectives * np.inner(self.range_n_features, x) - 1) ** 2

    def jac_f(self, x):
        x = self._x(x)
        i = self.range_n_objectives[:, None]
        return 2 * i * self.range_n_features * (i * np.inner(self.range_n_features, x) - 1)


def match_native_multi(f, g, jac_f, prox_wsum_g):
    """The built-in multi-objective Problem whose four bound methods these are, else None."""
    owner = getattr(f, "__self__", None)
    if not isinstance(owner, (_BuiltinProblem, _HostProblem)):
        return None
    for cb, name in zip((f, g, jac_f, prox_wsum_g), ("f", "g", "jac_f", "prox_wsum_g")):
        if getattr(cb, "__self__", None) is not owner:
            return None
        if getattr(cb, "__func__", None) is not getattr(type(owner), name):
            return None
    return owner
