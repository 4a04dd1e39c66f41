"""LevenbergMarquardtOptimizer mirror over the C ABI (include/lmgpu.h).

Same names / argument meaning / error behaviour as the reference interface it stands in for:
  LevenbergMarquardtParams       gtsam/nonlinear/LevenbergMarquardtParams.h:35-157
  LevenbergMarquardtOptimizer    gtsam/nonlinear/LevenbergMarquardtOptimizer.h  (iterate :103, linearize :113, lambda, getInnerIterations)
  NonlinearOptimizer             gtsam/nonlinear/NonlinearOptimizer.h (optimize :98, error, iterations, values, solve :129)
All numerics run in liblmgpu.so on the GPU; this file only marshals arrays.
"""
from __future__ import annotations

import ctypes as ct

import numpy as np

from . import _lib
from .graph import (CAM_BUNDLER, F_PRIOR_CAM, F_SFM, FACTOR_ARITY, N_UNIT, VAR_DIM, VAR_STORE, VAR_STORE_DEV, NonlinearFactorGraph, Ordering,
                    Values)


class LevenbergMarquardtParams:
    def __init__(self):
        self.ordering = None
        LevenbergMarquardtParams.SetLegacyDefaults(self)
        self.errorTol = 0.0
        self.minDiagonal = 1e-6
        self.maxDiagonal = 1e32

    @staticmethod
    def SetLegacyDefaults(p):  # LevenbergMarquardtParams.h:69-82
        p.maxIterations = 100
        p.relativeErrorTol = 1e-5
        p.absoluteErrorTol = 1e-5
        p.lambdaInitial = 1e-5
        p.lambdaFactor = 10.0
        p.lambdaUpperBound = 1e5
        p.lambdaLowerBound = 0.0
        p.minModelFidelity = 1e-3
        p.diagonalDamping = False
        p.useFixedLambdaFactor = True

    @staticmethod
    def SetCeresDefaults(p):  # LevenbergMarquardtParams.h:85-98
        p.maxIterations = 50
        p.absoluteErrorTol = 0.0
        p.relativeErrorTol = 1e-6
        p.lambdaUpperBound = 1e32
        p.lambdaLowerBound = 1e-16
        p.lambdaInitial = 1e-4
        p.lambdaFactor = 2.0
        p.minModelFidelity = 1e-3
        p.diagonalDamping = True
        p.useFixedLambdaFactor = False

    @staticmethod
    def LegacyDefaults():
        return LevenbergMarquardtParams()

    @staticmethod
    def CeresDefaults():
        p = LevenbergMarquardtParams()
        LevenbergMarquardtParams.SetCeresDefaults(p)
        return p

    def _c(self):
        return _lib.lmgpu_lm_params(int(self.maxIterations), self.relativeErrorTol, self.absoluteErrorTol, self.errorTol, self.lambdaInitial,
                                    self.lambdaFactor, self.lambdaUpperBound, self.lambdaLowerBound, self.minModelFidelity,
                                    int(bool(self.diagonalDamping)), int(bool(self.useFixedLambdaFactor)), self.minDiagonal, self.maxDiagonal)


def _dp(a):
    return a.ctypes.data_as(ct.POINTER(ct.c_double))


def _ip(a):
    return a.ctypes.data_as(ct.POINTER(ct.c_int32))


class LevenbergMarquardtOptimizer:
    """LevenbergMarquardtOptimizer(graph, initialValues, ordering, params).

    `ordering` (an Ordering = list of keys in elimination order) is a boundary input, as it is for the
    reference's hot loop (computed once at construction, LevenbergMarquardtParams.h:112-117).
    `device=-1` builds a structure-only handle (symbolic analysis, no compute)."""

    def __init__(self, graph: NonlinearFactorGraph, initialValues: Values, ordering=None, params: LevenbergMarquardtParams | None = None,
                 device: int = 0, rank: int = 0, world_size: int = 1, comm_id: bytes | None = None,
                 local_group=None, split_root: bool = False):
        self.params = params or LevenbergMarquardtParams()
        ordering = ordering if ordering is not None else self.params.ordering
        if ordering is None:
            raise ValueError("an elimination Ordering is required (Ordering.Schur / Ordering.Natural, or COLAMD/METIS from the caller)")
        self.graph, self.ordering = graph, Ordering(ordering)
        self.lib = _lib.load(test_hooks=local_group is not None)  # the in-process communicator lives in liblmgpu_test.so only
        self._h = ct.c_void_p()
        cfg = _lib.lmgpu_config(device, rank, world_size, 1 if split_root else 0)
        self._check(self.lib.lmgpu_create(ct.byref(cfg), ct.byref(self._h)))
        self.device = device
        # variables in elimination order
        self._keys = np.array(self.ordering, dtype=np.uint64)
        if len(set(self.ordering)) != len(self.ordering):
            raise ValueError("ordering has duplicate keys")
        self._slot = {int(k): i for i, k in enumerate(self.ordering)}
        self._types = np.array([initialValues.type(k) for k in self.ordering], dtype=np.int32)
        self._check(self.lib.lmgpu_set_variables(self._h, len(self.ordering), self._keys.ctypes.data_as(ct.POINTER(ct.c_uint64)), _ip(self._types)))
        self._template = initialValues.copy()
        order_sorted = np.argsort(self._keys, kind="stable")
        keys_sorted = self._keys[order_sorted]
        for ftype, kind, gi, keys, meas, noise, models in graph.buckets():
            ar = FACTOR_ARITY[ftype]
            pos = np.searchsorted(keys_sorted, keys.reshape(-1))
            if (pos >= len(keys_sorted)).any() or (keys_sorted[np.minimum(pos, len(keys_sorted) - 1)] != keys.reshape(-1)).any():
                raise KeyError("a factor references a key that is not in the ordering")
            slots = order_sorted[pos].astype(np.int32).reshape(-1, ar)
            meas = meas.copy()
            if ftype == F_SFM:
                # fold Cal3Bundler's constant principal point into z (see include/lmgpu.h, CAM_BUNDLER)
                ucam, inv = np.unique(keys[:, 0], return_inverse=True)
                uv = np.array([initialValues.at(k)[15:17] for k in ucam])[inv]
                meas = meas - uv
            if ftype == F_PRIOR_CAM:
                meas = np.ascontiguousarray(meas[:, :15])
            meas = np.ascontiguousarray(meas, dtype=np.float64)
            gi32 = np.ascontiguousarray(gi, dtype=np.int32)
            slots = np.ascontiguousarray(slots)
            nptr = _dp(np.ascontiguousarray(noise)) if kind != N_UNIT else None
            self._check(self.lib.lmgpu_add_factor_bucket_robust(self._h, ftype, len(gi32), _ip(gi32), _ip(slots), _dp(meas), kind, nptr,
                                                                models[0].robust_kind, models[0].robust_k))
        self._check(self.lib.lmgpu_finalize_structure(self._h))
        self._ntot = self.lib.lmgpu_total_dim(self._h)
        self._nstore = self.lib.lmgpu_total_store(self._h)
        self._voff = np.concatenate([[0], np.cumsum([VAR_STORE_DEV[t] for t in self._types])]).astype(np.int64)
        self._xoff = np.concatenate([[0], np.cumsum([VAR_DIM[t] for t in self._types])]).astype(np.int64)
        self.state = _lib.lmgpu_lm_state()
        if device >= 0:
            if world_size > 1 or split_root:
                if local_group is not None:  # in-process communicator (tests): one thread per rank
                    self._check(self.lib.lmgpu_comm_init_local(self._h, local_group))
                elif comm_id is None:
                    raise ValueError("world_size > 1 needs comm_id (lmgpu_comm_unique_id bytes broadcast from rank 0)")
                else:
                    self.comm_init(comm_id)
            self.set_values(initialValues)
            cp = self.params._c()
            self._check(self.lib.lmgpu_lm_init(self._h, ct.byref(cp), ct.byref(self.state)))

    # ------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc == _lib.LMGPU_OK:
            return
        if rc == _lib.LMGPU_INDETERMINATE:
            raise _lib.IndeterminantLinearSystemException(self.lib.lmgpu_last_failed_slot(self._h))
        msg = self.lib.lmgpu_last_error(self._h)
        raise _lib.LmgpuError(f"lmgpu status {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.lmgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _pack(self, values: Values):
        out = np.empty(self._nstore)
        for i, k in enumerate(self.ordering):
            n = VAR_STORE_DEV[self._types[i]]
            out[self._voff[i]:self._voff[i] + n] = values.at(k)[:n]
        return out

    def set_values(self, values: Values):
        packed = self._pack(values)
        self._check(self.lib.lmgpu_set_values(self._h, _dp(packed)))

    # ------------------------------------------------------------ NonlinearOptimizer interface
    def values(self) -> Values:
        packed = np.empty(self._nstore)
        self._check(self.lib.lmgpu_get_values(self._h, _dp(packed)))
        out = self._template.copy()
        for i, k in enumerate(self.ordering):
            n = VAR_STORE_DEV[self._types[i]]
            v = out.at(k).copy()
            v[:n] = packed[self._voff[i]:self._voff[i] + n]
            out.update(k, v)
        return out

    def error(self) -> float:
        return self.state.error

    def iterations(self) -> int:
        return self.state.iterations

    def lambda_(self) -> float:
        return self.state.lambda_

    def getInnerIterations(self) -> int:
        return self.state.totalNumberInnerIterations

    def iterate(self):
        """one outer iteration; returns the linearized graph like the reference (NonlinearOptimizer.h:136,
        LevenbergMarquardtOptimizer.cpp:281,307): the whitened Jacobians of the linearization this iteration solved, fetched from
        the device when first read"""
        cp = self.params._c()
        self._check(self.lib.lmgpu_iterate(self._h, ct.byref(cp), ct.byref(self.state)))
        return self._returned_linear_graph()

    def _returned_linear_graph(self):
        self._lin_generation = getattr(self, "_lin_generation", 0) + 1
        return GaussianFactorGraph(self, self._lin_generation)

    def linear_graph(self):
        """the linearization currently on the device as a GaussianFactorGraph (one copy per factor bucket)"""
        return GaussianFactorGraph(self, getattr(self, "_lin_generation", 0))

    def optimize(self) -> Values:
        cp = self.params._c()
        self._check(self.lib.lmgpu_optimize(self._h, ct.byref(cp), ct.byref(self.state)))
        self._lin_generation = getattr(self, "_lin_generation", 0) + 1
        return self.values()

    def timings(self):
        t = _lib.lmgpu_timings()
        self._check(self.lib.lmgpu_get_timings(self._h, ct.byref(t)))
        return {f: getattr(t, f) for f, _ in t._fields_}

    def save_values(self):
        self._check(self.lib.lmgpu_save_values(self._h))

    def restore_values(self, state=None):
        """restore the device-side snapshot; optionally also the LM state (a copy of a previous `opt.state`)"""
        self._check(self.lib.lmgpu_restore_values(self._h))
        if state is not None:
            ct.memmove(ct.byref(self.state), ct.byref(state), ct.sizeof(_lib.lmgpu_lm_state))

    def copy_state(self):
        s = _lib.lmgpu_lm_state()
        ct.memmove(ct.byref(s), ct.byref(self.state), ct.sizeof(_lib.lmgpu_lm_state))
        return s

    def set_kernel_timing(self, on=True):
        """True / 1: every category; 2: only the two roofline kernels (linearize, chain); False: off"""
        self._check(self.lib.lmgpu_set_kernel_timing(self._h, int(on)))

    def kernel_times(self):
        """{category: dict(ms, work, launches)} accumulated since set_kernel_timing(True)"""
        n = len(_lib.KT_NAMES)
        ms, work, cnt = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.int64)
        self._check(self.lib.lmgpu_get_kernel_times(self._h, _dp(ms), _dp(work), cnt.ctypes.data_as(ct.POINTER(ct.c_int64))))
        return {name: dict(ms=float(ms[i]), work=float(work[i]), launches=int(cnt[i])) for i, name in enumerate(_lib.KT_NAMES)}

    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclUniqueId bytes (rank 0 creates it; the caller broadcasts it to the other ranks)"""
        buf = ct.create_string_buffer(128)
        rc = _lib.load().lmgpu_comm_unique_id(buf)
        if rc != 0:
            raise _lib.LmgpuError("lmgpu_comm_unique_id failed")
        return buf.raw

    def comm_init(self, id128: bytes):
        self._check(self.lib.lmgpu_comm_init(self._h, id128))

    # ------------------------------------------------------------ piecewise hot path (tryLambda's calls)
    def graph_error(self) -> float:
        e = ct.c_double()
        self._check(self.lib.lmgpu_error(self._h, ct.byref(e)))
        return e.value

    def linearize(self):
        """LevenbergMarquardtOptimizer::linearize() (LevenbergMarquardtOptimizer.h:113): returns the linear graph (fetched on first read)"""
        self._check(self.lib.lmgpu_linearize(self._h))
        return self._returned_linear_graph()

    def solve(self, lam, diagonal_damping=False, min_diag=1e-6, max_diag=1e32):
        """returns (delta by key dict, packed delta in slot order, linear error at 0, linear error at delta)"""
        d = np.empty(self._ntot)
        e0, e1 = ct.c_double(), ct.c_double()
        self._check(self.lib.lmgpu_solve(self._h, lam, int(diagonal_damping), min_diag, max_diag, _dp(d), ct.byref(e0), ct.byref(e1)))
        return self.delta_by_key(d), d, e0.value, e1.value

    def delta_by_key(self, packed):
        return {int(k): packed[self._xoff[i]:self._xoff[i + 1]].copy() for i, k in enumerate(self.ordering)}

    def retract(self, packed_delta=None):
        p = None if packed_delta is None else _dp(np.ascontiguousarray(packed_delta, dtype=np.float64))
        self._check(self.lib.lmgpu_retract(self._h, p))

    def hessian_diagonal(self):
        d = np.empty(self._ntot)
        self._check(self.lib.lmgpu_hessian_diagonal(self._h, _dp(d)))
        return self.delta_by_key(d)

    # ------------------------------------------------------------ parity taps
    def jacobian(self, graph_index):
        r, c = ct.c_int32(), ct.c_int32()
        self._check(self.lib.lmgpu_get_jacobian(self._h, graph_index, None, ct.byref(r), ct.byref(c)))
        out = np.empty(r.value * c.value)
        self._check(self.lib.lmgpu_get_jacobian(self._h, graph_index, _dp(out), ct.byref(r), ct.byref(c)))
        return out.reshape(c.value, r.value).T.copy()  # column-major -> (rows, cols)

    def num_fronts(self):
        return self.lib.lmgpu_num_fronts(self._h)

    def front_info(self, i):
        info = np.zeros(8, dtype=np.int32)
        self._check(self.lib.lmgpu_front_info(self._h, i, _ip(info)))
        return dict(n_keys=int(info[0]), n_frontal_keys=int(info[1]), nf=int(info[2]), n=int(info[3]), parent=int(info[4]), cls=int(info[5]),
                    owner=int(info[6]), level=int(info[7]))

    def front(self, i, numeric=True):
        """(keys in Scatter order, [R S d] as (nf, n) array or None)"""
        fi = self.front_info(i)
        slots = np.zeros(fi["n_keys"], dtype=np.int32)
        rsd = np.empty(fi["nf"] * fi["n"]) if numeric else None
        self._check(self.lib.lmgpu_get_front(self._h, i, _ip(slots), _dp(rsd) if numeric else None))
        keys = [int(self._keys[s]) for s in slots]
        return keys, (rsd.reshape(fi["n"], fi["nf"]).T.copy() if numeric else None)


class JacobianFactor:
    """the part of gtsam/linear/JacobianFactor.h the returned linear graph is read through: keys(), getA(i) / jacobian(), getb(),
    error(x) = 0.5 ||A x - b||^2 (JacobianFactor.cpp:509-514); unit noise model (already whitened)"""

    def __init__(self, keys, dims, Ab):
        self._keys, self._dims, self._Ab = list(keys), list(dims), Ab
        self._off = np.concatenate([[0], np.cumsum(dims)]).astype(int)

    def keys(self):
        return list(self._keys)

    def getA(self, i):
        return self._Ab[:, self._off[i]:self._off[i + 1]].copy()

    def getb(self):
        return self._Ab[:, -1].copy()

    def jacobian(self):
        return self._Ab[:, :-1].copy(), self.getb()

    def augmentedJacobian(self):
        return self._Ab.copy()

    def error(self, x):
        r = -self._Ab[:, -1]
        for i, k in enumerate(self._keys):
            r = r + self._Ab[:, self._off[i]:self._off[i + 1]] @ np.asarray(x[k], dtype=float)
        return 0.5 * float(r @ r)


class GaussianFactorGraph:
    """What iterate() / linearize() return: the linearization on the device as a list of JacobianFactors, index-preserving like
    NonlinearFactorGraph::linearize (NonlinearFactorGraph.cpp:239-278; a slot without a factor is None).  The numbers are fetched at the
    first read (lmgpu_get_jacobians: one copy per factor bucket); a graph that is first read after the optimizer has linearized again
    refuses — its linearization is no longer on the device."""

    def __init__(self, opt, generation):
        self._opt, self._generation, self._factors = opt, generation, None

    def _fetch(self):
        if self._factors is not None:
            return
        o = self._opt
        if getattr(o, "_lin_generation", 0) != self._generation:
            raise _lib.LmgpuError("this linear graph was not read before the optimizer linearized again")
        n = ct.c_int32()
        o._check(o.lib.lmgpu_get_jacobians(o._h, ct.byref(n), None, None, None, None, None))
        gi, rows, cols = (np.zeros(n.value, dtype=np.int32) for _ in range(3))
        off = np.zeros(n.value + 1, dtype=np.int64)
        o._check(o.lib.lmgpu_get_jacobians(o._h, ct.byref(n), _ip(gi), _ip(rows), _ip(cols), off.ctypes.data_as(ct.POINTER(ct.c_int64)), None))
        out = np.empty(int(off[-1]))
        o._check(o.lib.lmgpu_get_jacobians(o._h, ct.byref(n), None, None, None, None, _dp(out)))
        fac = [None] * o.graph.size()
        fkeys = o.graph.factor_keys_in_graph_order()
        for i in range(n.value):
            Ab = out[off[i]:off[i + 1]].reshape(cols[i], rows[i]).T  # column-major -> (rows, cols)
            keys = fkeys[int(gi[i])]
            dims = [VAR_DIM[o._types[o._slot[int(k)]]] for k in keys]
            fac[int(gi[i])] = JacobianFactor(keys, dims, Ab)
        self._factors = fac

    def size(self):
        self._fetch()
        return len(self._factors)

    __len__ = size

    def at(self, i):
        self._fetch()
        return self._factors[i]

    __getitem__ = at

    def error(self, x):
        """GaussianFactorGraph::error (GaussianFactorGraph.cpp:71-78); x: {key: vector}"""
        self._fetch()
        return sum(f.error(x) for f in self._factors if f is not None)

    def gradientAtZero(self):
        """GaussianFactorGraph::gradientAtZero (GaussianFactorGraph.cpp:357-367): {key: -sum A_k^T b}"""
        self._fetch()
        g = {}
        for f in self._factors:
            if f is None:
                continue
            b = f.getb()
            for i, k in enumerate(f.keys()):
                g[k] = g.get(k, 0.0) - f.getA(i).T @ b
        return g


class GaussNewtonParams(LevenbergMarquardtParams):
    """NonlinearOptimizerParams defaults (gtsam/nonlinear/NonlinearOptimizerParams.h: maxIterations 100, relativeErrorTol 1e-5,
    absoluteErrorTol 1e-5, errorTol 0); the LM-only fields of the shared C struct are ignored by the Gauss-Newton entry points."""

    def __init__(self):
        super().__init__()


class GaussNewtonOptimizer(LevenbergMarquardtOptimizer):
    """gtsam/nonlinear/GaussNewtonOptimizer.h:38-91 on the same device-resident graph and kernels: iterate() = linearize,
    solve the undamped system, retract, new error (GaussNewtonOptimizer.cpp:44-66); optimize() = defaultOptimize."""

    def __init__(self, graph, initialValues, ordering=None, params=None, **kw):
        super().__init__(graph, initialValues, ordering, params or GaussNewtonParams(), **kw)

    def iterate(self):
        self._check(self.lib.lmgpu_gn_iterate(self._h, ct.byref(self.state)))
        return self._returned_linear_graph()  # GaussNewtonOptimizer.cpp:49,65

    def optimize(self) -> Values:
        cp = self.params._c()
        self._check(self.lib.lmgpu_gn_optimize(self._h, ct.byref(cp), ct.byref(self.state)))
        self._lin_generation = getattr(self, "_lin_generation", 0) + 1
        return self.values()


class DoglegParams(LevenbergMarquardtParams):
    """gtsam/nonlinear/DoglegOptimizer.h:33-63: NonlinearOptimizerParams + deltaInitial (default 1.0)"""

    def __init__(self):
        super().__init__()
        self.deltaInitial = 1.0


class DoglegOptimizer(LevenbergMarquardtOptimizer):
    """gtsam/nonlinear/DoglegOptimizer.h:69-133 (multifrontal elimination) on the device-resident Bayes tree."""

    def __init__(self, graph, initialValues, ordering=None, params=None, **kw):
        params = params or DoglegParams()
        super().__init__(graph, initialValues, ordering, params, **kw)
        self.state.lambda_ = float(getattr(params, "deltaInitial", 1.0))

    def getDelta(self) -> float:
        return self.state.lambda_

    def iterate(self):
        self._check(self.lib.lmgpu_dl_iterate(self._h, ct.byref(self.state)))
        return self._returned_linear_graph()  # DoglegOptimizer.cpp:87,122

    def optimize(self) -> Values:
        cp = self.params._c()
        self._check(self.lib.lmgpu_dl_optimize(self._h, ct.byref(cp), ct.byref(self.state)))
        self._lin_generation = getattr(self, "_lin_generation", 0) + 1
        return self.values()


class Marginals:
    """Marginals(graph, solution, ordering): marginal covariances of single variables at `solution`
    (gtsam/nonlinear/Marginals.h; constructor Marginals.cpp:28-43: linearize the graph at the solution and eliminate it;
    marginalCovariance :124-127, marginalInformation :109-121).  Cholesky factorization only (the reference's default).
    The elimination ordering is a boundary input as for the optimizers (the reference's default here is COLAMD; the result
    does not depend on it).  Raises IndeterminantLinearSystemException-like LmgpuError where the reference throws."""

    def __init__(self, graph: NonlinearFactorGraph, solution: Values, ordering, device: int = 0):
        self._opt = LevenbergMarquardtOptimizer(graph, solution, ordering, LevenbergMarquardtParams(), device=device)

    def marginalCovariance(self, key) -> np.ndarray:
        o = self._opt
        slot = o._slot[int(key)]
        d = int(o._xoff[slot + 1] - o._xoff[slot])
        out = np.zeros((d, d))
        o._check(o.lib.lmgpu_marginal_covariance(o._h, slot, _dp(out)))
        return out

    def marginalInformation(self, key) -> np.ndarray:
        return np.linalg.inv(self.marginalCovariance(key))

    def jointMarginalCovariance(self, keys) -> "JointMarginal":
        """Marginals::jointMarginalCovariance (Marginals.cpp:130-137): blocks ordered by Key like JointMarginal::fullMatrix"""
        o = self._opt
        keys = sorted(int(k) for k in keys)
        slots = np.array([o._slot[k] for k in keys], dtype=np.int32)
        dims = [int(o._xoff[s + 1] - o._xoff[s]) for s in slots]
        out = np.zeros((sum(dims), sum(dims)))
        o._check(o.lib.lmgpu_joint_marginal_covariance(o._h, len(slots), _ip(slots), _dp(out)))
        return JointMarginal(keys, dims, out)

    def jointMarginalInformation(self, keys) -> "JointMarginal":
        j = self.jointMarginalCovariance(keys)
        return JointMarginal(j.keys, j.dims, np.linalg.inv(j.fullMatrix()))

    def close(self):
        self._opt.close()


class JointMarginal:
    """JointMarginal (gtsam/nonlinear/Marginals.h:141-191): at(iVariable, jVariable) = block (i, j); fullMatrix() with the
    blocks ordered by Key"""

    def __init__(self, keys, dims, full):
        self.keys, self.dims, self._full = list(keys), list(dims), full
        off = np.concatenate([[0], np.cumsum(dims)])
        self._range = {k: (int(off[i]), int(off[i + 1])) for i, k in enumerate(keys)}

    def at(self, iVariable, jVariable) -> np.ndarray:
        (a0, a1), (b0, b1) = self._range[int(iVariable)], self._range[int(jVariable)]
        return self._full[a0:a1, b0:b1].copy()

    __call__ = at

    def fullMatrix(self) -> np.ndarray:
        return self._full.copy()
