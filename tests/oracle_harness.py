"""TEST INFRASTRUCTURE: ctypes harness over oracle/liblm_oracle.so (the CPU restatement of the reference's
LM path) and oracle/_ref/*.so (CCOLAMD / METIS compiled from the reference's vendored C sources).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes as ct
import os

import numpy as np

from gtsam_personal_amd.graph import FACTOR_ARITY, FACTOR_MEAS, N_DIAG, N_GAUSS, N_ISO, N_UNIT, VAR_DIM, VAR_STORE, NonlinearFactorGraph, Values

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liblm_oracle.so")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")

_D = ct.POINTER(ct.c_double)
_I = ct.POINTER(ct.c_int)
_U = ct.POINTER(ct.c_uint64)


class orc_lm_params(ct.Structure):
    _fields_ = [
        ("maxIterations", ct.c_int),
        ("relativeErrorTol", ct.c_double), ("absoluteErrorTol", ct.c_double), ("errorTol", ct.c_double),
        ("lambdaInitial", ct.c_double), ("lambdaFactor", ct.c_double), ("lambdaUpperBound", ct.c_double), ("lambdaLowerBound", ct.c_double),
        ("minModelFidelity", ct.c_double),
        ("diagonalDamping", ct.c_int), ("useFixedLambdaFactor", ct.c_int),
        ("minDiagonal", ct.c_double), ("maxDiagonal", ct.c_double),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ct.CDLL(ORACLE_SO)
        L.orc_create.restype = ct.c_void_p
        L.orc_linear_create.restype = ct.c_void_p
        L.orc_error.restype = ct.c_double
        for name in ("orc_destroy", "orc_linear_destroy"):
            getattr(L, name).argtypes = [ct.c_void_p]
        L.orc_add_variable.argtypes = [ct.c_void_p, ct.c_uint64, ct.c_int, _D]
        L.orc_add_factor.argtypes = [ct.c_void_p, ct.c_int, _U, _D, ct.c_int, _D]
        L.orc_set_ordering.argtypes = [ct.c_void_p, ct.c_int, _U]
        L.orc_set_factor_robust.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_double]
        L.orc_robust.argtypes = [ct.c_int, ct.c_double, ct.c_double, _D]
        L.orc_get_values.argtypes = [ct.c_void_p, _D]
        L.orc_marginal_covariance.argtypes = [ct.c_void_p, ct.c_uint64, _D]
        L.orc_error.argtypes = [ct.c_void_p]
        L.orc_linearize.argtypes = [ct.c_void_p]
        L.orc_get_jacobian.argtypes = [ct.c_void_p, ct.c_int, _D, _I, _I]
        L.orc_solve.argtypes = [ct.c_void_p, ct.c_double, ct.c_int, ct.c_double, ct.c_double, _D, _D, _D]
        L.orc_hessian_diagonal.argtypes = [ct.c_void_p, _D]
        L.orc_retract.argtypes = [ct.c_void_p, _D]
        L.orc_num_cliques.argtypes = [ct.c_void_p]
        L.orc_clique_info.argtypes = [ct.c_void_p, ct.c_int, _I]
        L.orc_clique_get.argtypes = [ct.c_void_p, ct.c_int, _U, _D]
        for name in ("orc_lm_init", "orc_lm_iterate", "orc_lm_optimize", "orc_gn_optimize"):
            getattr(L, name).argtypes = [ct.c_void_p, ct.POINTER(orc_lm_params)]
        L.orc_gn_iterate.argtypes = [ct.c_void_p]
        L.orc_dl_init.argtypes = [ct.c_void_p, ct.c_double]
        L.orc_dl_iterate.argtypes = [ct.c_void_p]
        L.orc_dl_optimize.argtypes = [ct.c_void_p, ct.POINTER(orc_lm_params)]
        L.orc_get_delta.argtypes = [ct.c_void_p, _D]
        L.orc_dl_points.argtypes = [ct.c_void_p, _D, _D]
        L.orc_dogleg_point.argtypes = [ct.c_int, _D, _D, ct.c_double, _D]
        L.orc_lm_state.argtypes = [ct.c_void_p, _D]
        L.orc_lm_trace_len.argtypes = [ct.c_void_p]
        L.orc_lm_trace.argtypes = [ct.c_void_p, _D]
        L.orc_timings.argtypes = [ct.c_void_p, _D]
        L.orc_num_variables.argtypes = [ct.c_void_p]
        L.orc_num_factors.argtypes = [ct.c_void_p]
        L.orc_total_dim.argtypes = [ct.c_void_p]
        L.orc_cholesky_partial.argtypes = [_D, ct.c_int, ct.c_int]
        L.orc_linear_add_jacobian.argtypes = [ct.c_void_p, ct.c_int, _U, _I, ct.c_int, _D, _D, _D]
        L.orc_linear_add_hessian.argtypes = [ct.c_void_p, ct.c_int, _U, _I, _D]
        L.orc_linear_eliminate_dense.argtypes = [ct.c_void_p, ct.c_int, _U, _I, _U, _I, _I, _D, _D]
        L.orc_linear_optimize.argtypes = [ct.c_void_p, ct.c_int, _U, _D]
        L.orc_linear_num_cliques.argtypes = [ct.c_void_p]
        L.orc_linear_clique_info.argtypes = [ct.c_void_p, ct.c_int, _I]
        L.orc_linear_clique_get.argtypes = [ct.c_void_p, ct.c_int, _U, _D]
        L.orc_cal3bundler_uncalibrate.argtypes = [_D, ct.c_double, ct.c_double, _D, _D, _D]
        L.orc_pose3_expmap.argtypes = [_D, _D]
        L.orc_pose3_logmap.argtypes = [_D, _D]
        L.orc_rot3_expmap.argtypes = [_D, _D]
        L.orc_rot3_logmap.argtypes = [_D, _D]
        L.orc_factor_evaluate.argtypes = [ct.c_void_p, ct.c_int, _D, _D, _D]
        L.orc_factor_evaluate3.argtypes = [ct.c_void_p, ct.c_int, _D, _D, _D, _D]
        L.orc_set_threads.argtypes = [ct.c_int]
        L.orc_set_dense_threads.argtypes = [ct.c_int]
        _lib = L
    return _lib


def set_threads(n):
    """threads of the oracle's all-cores leg (subtree-parallel elimination + parallel linearize, like the reference with TBB)"""
    return lib().orc_set_threads(int(n))


def set_dense_threads(n):
    """threads inside the blocked dense factorisation of a large front (1 = like the reference: Eigen's LLT is single-threaded
    whatever GTSAM's TBB setting)"""
    return lib().orc_set_dense_threads(int(n))


def dp(a):
    return a.ctypes.data_as(_D)


def up(a):
    return a.ctypes.data_as(_U)


def ip(a):
    return a.ctypes.data_as(_I)


def lm_params_c(p):
    return orc_lm_params(int(p.maxIterations), p.relativeErrorTol, p.absoluteErrorTol, p.errorTol, p.lambdaInitial, p.lambdaFactor,
                         p.lambdaUpperBound, p.lambdaLowerBound, p.minModelFidelity, int(bool(p.diagonalDamping)),
                         int(bool(p.useFixedLambdaFactor)), p.minDiagonal, p.maxDiagonal)


class OracleProblem:
    """the same (graph, values, ordering) fed to the CPU oracle"""

    def __init__(self, graph: NonlinearFactorGraph, values: Values, ordering):
        self.L = lib()
        self.h = ct.c_void_p(self.L.orc_create())
        self.keys = values.keys()
        self.types = [values.type(k) for k in self.keys]
        for k in self.keys:
            v = np.ascontiguousarray(values.at(k), dtype=np.float64)
            assert self.L.orc_add_variable(self.h, k, values.type(k), dp(v)) == 0
        # factors in graph order
        n = graph.size()
        rec = [None] * n
        for ftype, kind, gi, keys, meas, noise, models in graph.buckets():
            for i, g in enumerate(gi.tolist()):
                rec[g] = (ftype, keys[i], meas[i], models[i])
        for ftype, keys, meas, model in rec:
            kk = np.zeros(3, dtype=np.uint64)
            kk[:FACTOR_ARITY[ftype]] = keys
            m = np.ascontiguousarray(meas, dtype=np.float64)
            if model.kind == N_UNIT:
                nd = None
            elif model.kind == N_ISO:
                nd = dp(np.array([float(model.data)]))
            else:
                nd = dp(np.ascontiguousarray(model.data, dtype=np.float64).reshape(-1))
            assert self.L.orc_add_factor(self.h, ftype, up(kk), dp(m), model.kind, nd) == 0
            if getattr(model, "robust_kind", 0):
                assert self.L.orc_set_factor_robust(self.h, self.L.orc_num_factors(self.h) - 1, model.robust_kind, model.robust_k) == 0
        o = np.array(list(ordering), dtype=np.uint64)
        self.L.orc_set_ordering(self.h, len(o), up(o))
        self.ntot = self.L.orc_total_dim(self.h)
        self.xoff = np.concatenate([[0], np.cumsum([VAR_DIM[t] for t in self.types])]).astype(int)

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def error(self):
        return self.L.orc_error(self.h)

    def linearize(self):
        self.L.orc_linearize(self.h)

    def jacobian(self, i):
        r, c = ct.c_int(), ct.c_int()
        self.L.orc_get_jacobian(self.h, i, None, ct.byref(r), ct.byref(c))
        out = np.empty(r.value * c.value)
        self.L.orc_get_jacobian(self.h, i, dp(out), ct.byref(r), ct.byref(c))
        return out.reshape(c.value, r.value).T.copy()

    def solve(self, lam, diagonal=False, min_diag=1e-6, max_diag=1e32):
        d = np.empty(self.ntot)
        e0, e1 = ct.c_double(), ct.c_double()
        rc = self.L.orc_solve(self.h, lam, int(diagonal), min_diag, max_diag, dp(d), ct.byref(e0), ct.byref(e1))
        return rc, self.by_key(d), e0.value, e1.value

    def by_key(self, packed):
        return {k: packed[self.xoff[i]:self.xoff[i + 1]].copy() for i, k in enumerate(self.keys)}

    def hessian_diagonal(self):
        d = np.empty(self.ntot)
        self.L.orc_hessian_diagonal(self.h, dp(d))
        return self.by_key(d)

    def retract(self, delta_by_key):
        d = np.concatenate([delta_by_key[k] for k in self.keys])
        self.L.orc_retract(self.h, dp(np.ascontiguousarray(d)))

    def values(self):
        tot = sum(VAR_STORE[t] for t in self.types)
        out = np.empty(tot)
        self.L.orc_get_values(self.h, dp(out))
        res, o = {}, 0
        for k, t in zip(self.keys, self.types):
            res[k] = out[o:o + VAR_STORE[t]].copy()
            o += VAR_STORE[t]
        return res

    def cliques(self):
        """[(keys, n_frontal_keys, RSd (nf, n), parent)] in post-order"""
        out = []
        for i in range(self.L.orc_num_cliques(self.h)):
            info = np.zeros(5, dtype=np.int32)
            self.L.orc_clique_info(self.h, i, ip(info))
            keys = np.zeros(info[0], dtype=np.uint64)
            rsd = np.empty(info[2] * info[3])
            self.L.orc_clique_get(self.h, i, up(keys), dp(rsd))
            out.append(([int(k) for k in keys], int(info[1]), rsd.reshape(info[3], info[2]).T.copy(), int(info[4])))
        return out

    def lm_init(self, params):
        c = lm_params_c(params)
        self.L.orc_lm_init(self.h, ct.byref(c))

    def lm_iterate(self, params):
        c = lm_params_c(params)
        self.L.orc_lm_iterate(self.h, ct.byref(c))

    def lm_optimize(self, params):
        c = lm_params_c(params)
        self.L.orc_lm_optimize(self.h, ct.byref(c))

    def gn_iterate(self):
        return self.L.orc_gn_iterate(self.h)

    def gn_optimize(self, params):
        c = lm_params_c(params)
        return self.L.orc_gn_optimize(self.h, ct.byref(c))

    def dl_points(self):
        xu, xn = np.empty(self.ntot), np.empty(self.ntot)
        assert self.L.orc_dl_points(self.h, dp(xu), dp(xn)) == 0
        return xu, xn

    def marginal_covariance(self, key, dim):
        """Marginals(graph, values).marginalCovariance(key) at the oracle's current values; None if the information matrix is singular"""
        out = np.empty((dim, dim))
        rc = self.L.orc_marginal_covariance(self.h, int(key), dp(out))
        assert rc in (0, 1), rc
        return out if rc == 0 else None

    def get_delta(self):
        d = np.empty(self.ntot)
        self.L.orc_get_delta(self.h, dp(d))
        return d

    def dl_init(self, delta_initial=1.0):
        return self.L.orc_dl_init(self.h, float(delta_initial))

    def dl_iterate(self):
        return self.L.orc_dl_iterate(self.h)

    def dl_optimize(self, params):
        c = lm_params_c(params)
        return self.L.orc_dl_optimize(self.h, ct.byref(c))

    def lm_state(self):
        s = np.empty(5)
        self.L.orc_lm_state(self.h, dp(s))
        return dict(error=s[0], lambda_=s[1], iterations=int(s[2]), inner=int(s[3]), factor=s[4])

    def lm_trace(self):
        n = self.L.orc_lm_trace_len(self.h)
        t = np.empty(n)
        self.L.orc_lm_trace(self.h, dp(t))
        return t.reshape(-1, 5)  # lambda, newError, modelFidelity, accepted, solved

    def timings(self):
        t = np.empty(3)
        self.L.orc_timings(self.h, dp(t))
        return dict(linearize_s=t[0], eliminate_s=t[1], backsub_s=t[2])


# ---------------------------------------------------------------- reference orderings (oracle/_ref)
def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "libccolamd_ref.so")) and os.path.exists(os.path.join(REF_DIR, "libmetis_ref.so"))


def variable_index(graph: NonlinearFactorGraph):
    """VariableIndex (gtsam/inference/VariableIndex-inl.h:27-49): sorted key -> ascending factor indices"""
    vi = {}
    for i, keys in enumerate(graph.factor_keys_in_graph_order()):
        for k in keys:
            vi.setdefault(k, []).append(i)
    return dict(sorted(vi.items()))


def colamd_from_index(vi, n_factors, cmember=None):
    """Ordering::ColamdConstrained (gtsam/inference/Ordering.cpp:50-125) on top of the reference's own ccolamd.c"""
    L = ct.CDLL(os.path.join(REF_DIR, "libccolamd_ref.so"))
    L.ccolamd_recommended.restype = ct.c_size_t
    L.ccolamd_recommended.argtypes = [ct.c_int, ct.c_int, ct.c_int]
    keys = list(vi.keys())
    nVars = len(keys)
    if nVars == 0:
        return []
    if nVars == 1:
        return keys
    nEntries = sum(len(v) for v in vi.values())
    Alen = L.ccolamd_recommended(nEntries, n_factors, nVars)
    A = np.zeros(Alen, dtype=np.int32)
    p = np.zeros(nVars + 1, dtype=np.int32)
    cnt = 0
    for idx, k in enumerate(keys):
        col = vi[k]
        A[cnt:cnt + len(col)] = col
        cnt += len(col)
        p[idx + 1] = cnt
    knobs = (ct.c_double * 20)()
    L.ccolamd_set_defaults(knobs)
    knobs[0] = -1  # CCOLAMD_DENSE_ROW
    knobs[1] = -1  # CCOLAMD_DENSE_COL
    stats = (ct.c_int * 20)()
    cm = np.zeros(nVars, dtype=np.int32) if cmember is None else np.asarray(cmember, dtype=np.int32)
    rv = L.ccolamd(ct.c_int(n_factors), ct.c_int(nVars), ct.c_int(Alen), ip(A), ip(p), knobs, stats, ip(cm))
    if rv != 1:
        raise RuntimeError(f"ccolamd failed with return value {rv}")
    return [keys[p[j]] for j in range(nVars)]


def colamd(graph: NonlinearFactorGraph):
    return colamd_from_index(variable_index(graph), graph.size())


def metis_from_adjacency(xadj, adj):
    L = ct.CDLL(os.path.join(REF_DIR, "libmetis_ref.so"))
    n = ct.c_int32(len(xadj) - 1)
    xadj = np.asarray(xadj, dtype=np.int32)
    adj = np.asarray(adj, dtype=np.int32)
    perm = np.zeros(n.value, dtype=np.int32)
    iperm = np.zeros(n.value, dtype=np.int32)
    rc = L.METIS_NodeND(ct.byref(n), ip(xadj), ip(adj), None, None, ip(perm), ip(iperm))
    if rc != 1:
        raise RuntimeError("METIS failed")
    return perm, iperm


def metis_index(factor_keys):
    """MetisIndex::augment (gtsam/inference/MetisIndex-inl.h:27-82): keys are numbered in order of first
    appearance; CSR adjacency rows in integer order (only keys that have a neighbour get a row, as in the reference)."""
    int_of, keys = {}, []
    for fk in factor_keys:
        for k in fk:
            if k not in int_of:
                int_of[k] = len(keys)
                keys.append(k)
    adjmap = {}
    for fk in factor_keys:
        for k1 in fk:
            for k2 in fk:
                if k1 != k2:
                    adjmap.setdefault(int_of[k1], set()).add(int_of[k2])
    xadj, adj = [0], []
    for i in sorted(adjmap):
        adj.extend(sorted(adjmap[i]))
        xadj.append(len(adj))
    return keys, xadj, adj


def metis(graph: NonlinearFactorGraph):
    return metis_from_factor_keys(graph.factor_keys_in_graph_order())


def metis_from_factor_keys(factor_keys):
    """Ordering::Metis (gtsam/inference/Ordering.cpp:211-256)"""
    keys, xadj, adj = metis_index(factor_keys)
    if len(keys) == 0:
        return []
    if len(keys) == 1:
        return keys
    perm, _ = metis_from_adjacency(xadj, adj)
    return [keys[i] for i in perm]


# ---------------------------------------------------------------- ISAM2 (oracle/isam2_oracle.hpp)
CCOLAMD_FN = ct.CFUNCTYPE(ct.c_int, ct.c_int, ct.c_int, _I, _I, _I, _I)


def ccolamd_csc(n_rows, n_cols, col_ptr, row_idx, cmember):
    """ccolamd of the reference (oracle/_ref) with GTSAM's knobs (gtsam/inference/Ordering.cpp:86-108): dense-row / dense-column
    detection off; returns the column permutation"""
    L = ct.CDLL(os.path.join(REF_DIR, "libccolamd_ref.so"))
    L.ccolamd_recommended.restype = ct.c_size_t
    L.ccolamd_recommended.argtypes = [ct.c_int, ct.c_int, ct.c_int]
    nnz = int(col_ptr[n_cols])
    Alen = L.ccolamd_recommended(nnz, n_rows, n_cols)
    A = np.zeros(Alen, dtype=np.int32)
    A[:nnz] = row_idx[:nnz]
    p = np.array(col_ptr[:n_cols + 1], dtype=np.int32)
    knobs = (ct.c_double * 20)()
    L.ccolamd_set_defaults(knobs)
    knobs[0] = -1
    knobs[1] = -1
    stats = (ct.c_int * 20)()
    cm = np.array(cmember[:n_cols], dtype=np.int32)
    rv = L.ccolamd(ct.c_int(n_rows), ct.c_int(n_cols), ct.c_int(Alen), ip(A), ip(p), knobs, stats, ip(cm))
    if rv != 1:
        raise RuntimeError(f"ccolamd failed with return value {rv}")
    return p[:n_cols].copy()


def _ccolamd_callback(n_rows, n_cols, col_ptr, row_idx, cmember, perm_out):
    try:
        cp = np.ctypeslib.as_array(col_ptr, shape=(n_cols + 1,))
        ri = np.ctypeslib.as_array(row_idx, shape=(max(1, int(cp[n_cols])),))
        cm = np.ctypeslib.as_array(cmember, shape=(n_cols,))
        perm = ccolamd_csc(n_rows, n_cols, cp, ri, cm)
        out = np.ctypeslib.as_array(perm_out, shape=(n_cols,))
        out[:] = perm
        return 1
    except Exception:  # noqa: BLE001 -- a Python exception must not unwind through the C caller
        import traceback
        traceback.print_exc()
        return 0


CCOLAMD_CALLBACK = CCOLAMD_FN(_ccolamd_callback)


class OracleISAM2:
    """ISAM2 (gtsam/nonlinear/ISAM2.h) restated on the CPU; Gauss-Newton optimisation params, Cholesky, COLAMD"""

    def __init__(self, relinearizeThreshold=0.1, relinearizeSkip=10, enableRelinearization=True, wildfireThreshold=0.001):
        self.L = lib()
        L = self.L
        L.orc_isam2_create.restype = ct.c_void_p
        L.orc_isam2_create.argtypes = [ct.c_double, ct.c_int, ct.c_int, ct.c_double, CCOLAMD_FN]
        L.orc_isam2_destroy.argtypes = [ct.c_void_p]
        L.orc_isam2_add_variable.argtypes = [ct.c_void_p, ct.c_uint64, ct.c_int, _D]
        L.orc_isam2_add_factor.argtypes = [ct.c_void_p, ct.c_int, _U, _D, ct.c_int, _D]
        L.orc_isam2_add_factor_robust.argtypes = [ct.c_void_p, ct.c_int, _U, _D, ct.c_int, _D, ct.c_int, ct.c_double]
        L.orc_isam2_update.argtypes = [ct.c_void_p, ct.c_int, _I]
        L.orc_isam2_set_thresholds.argtypes = [ct.c_void_p, ct.c_int, ct.c_char_p, _I, _D]
        L.orc_isam2_set_partial_check.argtypes = [ct.c_void_p, ct.c_int]
        L.orc_isam2_set_dogleg.argtypes = [ct.c_void_p, ct.c_double, ct.c_double, ct.c_int]
        L.orc_isam2_dogleg_delta.argtypes = [ct.c_void_p]
        L.orc_isam2_dogleg_delta.restype = ct.c_double
        L.orc_isam2_set_evaluate_error.argtypes = [ct.c_void_p, ct.c_int]
        L.orc_isam2_errors.argtypes = [ct.c_void_p, _D, _D]
        L.orc_isam2_error.argtypes = [ct.c_void_p, ct.c_int]
        L.orc_isam2_error.restype = ct.c_double
        L.orc_isam2_update_with.argtypes = [ct.c_void_p, ct.c_int, _U, ct.c_int, ct.c_int, _U, _I, ct.c_int, _U, ct.c_int, _U, ct.c_int, ct.c_int, _I]
        L.orc_isam2_unused_keys.argtypes = [ct.c_void_p, _U]
        L.orc_isam2_factor_exists.argtypes = [ct.c_void_p, ct.c_int]
        L.orc_isam2_num_factors.argtypes = [ct.c_void_p]
        L.orc_isam2_num_variables.argtypes = [ct.c_void_p]
        L.orc_isam2_values.argtypes = [ct.c_void_p, ct.c_int, _U, _I, _D]
        L.orc_isam2_delta.argtypes = [ct.c_void_p, _D]
        L.orc_isam2_snapshot.argtypes = [ct.c_void_p]
        L.orc_isam2_clique_info.argtypes = [ct.c_void_p, ct.c_int, _I]
        L.orc_isam2_clique_get.argtypes = [ct.c_void_p, ct.c_int, _U, _D]
        L.orc_isam2_set_find_unused_slots.argtypes = [ct.c_void_p, ct.c_int]
        L.orc_isam2_marginalize_leaves.argtypes = [ct.c_void_p, ct.c_int, _U, _U, _U, _I]
        L.orc_isam2_marginal_factor.argtypes = [ct.c_void_p, ct.c_int, _U, _I, _D]
        L.orc_isam2_fixed_variables.argtypes = [ct.c_void_p, _U]
        L.orc_isam2_graph_hessian.argtypes = [ct.c_void_p, _D]
        self.h = ct.c_void_p(L.orc_isam2_create(relinearizeThreshold, relinearizeSkip, int(enableRelinearization), wildfireThreshold,
                                                CCOLAMD_CALLBACK))

    def __del__(self):
        try:
            self.L.orc_isam2_destroy(self.h)
        except Exception:
            pass

    def set_relinearize_thresholds(self, thresholds):
        """ISAM2Params::relinearizeThreshold = FastMap<char, Vector>: {character: per-dof thresholds}; {} or None: the double again"""
        items = sorted((thresholds or {}).items())
        chrs = bytes(ord(c) if isinstance(c, str) else int(c) for c, _ in items)
        dims = np.asarray([len(v) for _, v in items], dtype=np.int32)
        vals = np.asarray([x for _, v in items for x in v], dtype=np.float64)
        self.L.orc_isam2_set_thresholds(self.h, len(items), chrs, ip(dims) if len(items) else None, dp(vals) if len(items) else None)

    def set_partial_relinearization_check(self, enable):
        self.L.orc_isam2_set_partial_check(self.h, int(bool(enable)))

    def set_dogleg(self, initialDelta=1.0, wildfireThreshold=1e-5, adaptationMode=0):
        """ISAM2Params::optimizationParams = ISAM2DoglegParams(...) (ISAM2Params.h:68-110); adaptationMode: 0 SEARCH_EACH_ITERATION,
        1 SEARCH_REDUCE_ONLY, 2 ONE_STEP_PER_ITERATION (DoglegOptimizerImpl.h:54-58).  Before the first update."""
        self.L.orc_isam2_set_dogleg(self.h, float(initialDelta), float(wildfireThreshold), int(adaptationMode))

    def doglegDelta(self):
        """the current trust-region radius (ISAM2::doglegDelta_)"""
        return float(self.L.orc_isam2_dogleg_delta(self.h))

    def set_evaluate_nonlinear_error(self, enable):
        self.L.orc_isam2_set_evaluate_error(self.h, int(bool(enable)))

    def errors(self):
        """(ISAM2Result::errorBefore, errorAfter) of the last update"""
        b, a = np.zeros(1), np.zeros(1)
        self.L.orc_isam2_errors(self.h, dp(b), dp(a))
        return float(b[0]), float(a[0])

    def error(self, which=0):
        """getFactorsUnsafe().error(calculateEstimate()) (which = 0) / at the linearization point (which = 2)"""
        return float(self.L.orc_isam2_error(self.h, int(which)))

    def update(self, newFactors: NonlinearFactorGraph = None, newTheta: Values = None, removeFactorIndices=(), constrainedKeys=None,
               noRelinKeys=None, extraReelimKeys=None, force_relinearize=False, forceFullSolve=False):
        """ISAM2::update(newFactors, newTheta, removeFactorIndices, constrainedKeys, noRelinKeys, extraReelimKeys, force_relinearize)
        (gtsam/nonlinear/ISAM2.h:146-186); returns dict(variablesRelinearized, variablesReeliminated, factorsRecalculated, cliques, batch)"""
        if newTheta is not None:
            for k in newTheta.keys():
                v = np.ascontiguousarray(newTheta.at(k), dtype=np.float64)
                assert self.L.orc_isam2_add_variable(self.h, k, newTheta.type(k), dp(v)) == 0
        if newFactors is not None and newFactors.size():
            rec = [None] * newFactors.size()
            for ftype, kind, gi, keys, meas, noise, models in newFactors.buckets():
                for i, g in enumerate(gi.tolist()):
                    rec[g] = (ftype, keys[i], meas[i], models[i])
            for ftype, keys, meas, model in rec:
                kk = np.zeros(3, dtype=np.uint64)
                kk[:FACTOR_ARITY[ftype]] = keys
                m = np.ascontiguousarray(meas, dtype=np.float64)
                if model.kind == N_UNIT:
                    nd = None
                elif model.kind == N_ISO:
                    nd = dp(np.array([float(model.data)]))
                else:
                    nd = dp(np.ascontiguousarray(model.data, dtype=np.float64).reshape(-1))
                if getattr(model, "robust_kind", 0):
                    assert self.L.orc_isam2_add_factor_robust(self.h, ftype, up(kk), dp(m), model.kind, nd, int(model.robust_kind), float(model.robust_k)) == 0
                else:
                    assert self.L.orc_isam2_add_factor(self.h, ftype, up(kk), dp(m), model.kind, nd) == 0
        res = np.zeros(5, dtype=np.int32)
        rm = np.asarray(list(removeFactorIndices), dtype=np.uint64)
        ck = np.asarray(sorted(constrainedKeys) if constrainedKeys else [], dtype=np.uint64)
        cg = np.asarray([constrainedKeys[int(k)] for k in ck], dtype=np.int32)
        nr = np.asarray(list(noRelinKeys or []), dtype=np.uint64)
        ex = np.asarray(list(extraReelimKeys or []), dtype=np.uint64)
        rc = self.L.orc_isam2_update_with(self.h, len(rm), up(rm), int(constrainedKeys is not None), len(ck), up(ck), ip(cg), len(nr), up(nr),
                                          len(ex), up(ex), int(force_relinearize), int(forceFullSolve), ip(res))
        assert rc == 0, rc
        return dict(variablesRelinearized=int(res[0]), variablesReeliminated=int(res[1]), factorsRecalculated=int(res[2]), cliques=int(res[3]),
                    batch=int(res[4]))

    def marginalCovariance(self, key):
        """ISAM2::marginalCovariance(key) (gtsam/nonlinear/ISAM2.h:253-257): the block of (sum over cliques [R S]^T [R S])^-1 -- the
        Bayes tree read as a Gaussian factor graph (GaussianBayesTree), dense; for test-sized problems"""
        lin = self._values(2)
        off, o = {}, 0
        for k in lin.keys():
            off[k] = o
            o += VAR_DIM[lin.type(k)]
        H = np.zeros((o, o))
        for keys, _, rsd, _ in self.cliques():
            cols = np.concatenate([np.arange(off[k], off[k] + VAR_DIM[lin.type(k)]) for k in keys])
            RS = rsd[:, :-1]
            H[np.ix_(cols, cols)] += RS.T @ RS
        d = VAR_DIM[lin.type(int(key))]
        return np.linalg.inv(H)[off[int(key)]:off[int(key)] + d, off[int(key)]:off[int(key)] + d]

    def set_find_unused_factor_slots(self, enable):
        """ISAM2Params::findUnusedFactorSlots"""
        self.L.orc_isam2_set_find_unused_slots(self.h, int(bool(enable)))

    def marginalizeLeaves(self, leafKeys):
        """ISAM2::marginalizeLeaves(leafKeys, &marginalFactorsIndices, &deletedFactorsIndices) (gtsam/nonlinear/ISAM2.h:198-222);
        returns (marginalFactorsIndices, deletedFactorsIndices).  RuntimeError when a key is not a leaf."""
        keys = np.asarray([int(k) for k in leafKeys], dtype=np.uint64)
        cap = self.num_factors() + 4 * len(keys) + 16
        mi, di, cnt = np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.uint64), np.zeros(2, dtype=np.int32)
        rc = self.L.orc_isam2_marginalize_leaves(self.h, len(keys), up(keys), up(mi), up(di), ip(cnt))
        if rc:
            raise RuntimeError("marginalizeLeaves: rc %d" % rc)
        return [int(i) for i in mi[:cnt[0]]], [int(i) for i in di[:cnt[1]]]

    def marginal_factor(self, i):
        """slot i as the LinearContainerFactor marginalizeLeaves left there: (keys, dims, augmented information matrix) or None"""
        n = self.L.orc_isam2_marginal_factor(self.h, int(i), None, None, None)
        if n < 0:
            return None
        keys, dims = np.zeros(max(n, 1), dtype=np.uint64), np.zeros(max(n, 1), dtype=np.int32)
        self.L.orc_isam2_marginal_factor(self.h, int(i), up(keys), ip(dims), None)
        N = int(dims[:n].sum()) + 1
        info = np.zeros(N * N)
        self.L.orc_isam2_marginal_factor(self.h, int(i), None, None, dp(info))
        return [int(k) for k in keys[:n]], [int(d) for d in dims[:n]], info.reshape(N, N).T.copy()

    def getFixedVariables(self):
        n = self.L.orc_isam2_fixed_variables(self.h, None)
        keys = np.zeros(max(n, 1), dtype=np.uint64)
        self.L.orc_isam2_fixed_variables(self.h, up(keys))
        return [int(k) for k in keys[:n]]

    def graph_augmented_hessian(self):
        """getFactorsUnsafe().linearize(getLinearizationPoint())->augmentedHessian(), variables ascending by key"""
        D = self.L.orc_isam2_graph_hessian(self.h, None)
        out = np.zeros((D + 1) * (D + 1))
        self.L.orc_isam2_graph_hessian(self.h, dp(out))
        return out.reshape(D + 1, D + 1)

    def tree_augmented_hessian(self):
        """GaussianFactorGraph(isam).augmentedHessian(): sum over cliques of [R S d]^T [R S d], variables ascending by key"""
        lin = self._values(2)
        off, o = {}, 0
        for k in lin.keys():
            off[k] = o
            o += VAR_DIM[lin.type(k)]
        H = np.zeros((o + 1, o + 1))
        for keys, _, rsd, _ in self.cliques():
            cols = np.concatenate([np.arange(off[k], off[k] + VAR_DIM[lin.type(k)]) for k in keys] + [np.array([o])]).astype(int)
            H[np.ix_(cols, cols)] += rsd.T @ rsd
        return H

    def unusedKeys(self):
        """ISAM2Result::unusedKeys of the last update"""
        n = self.L.orc_isam2_unused_keys(self.h, None)
        keys = np.zeros(max(n, 1), dtype=np.uint64)
        self.L.orc_isam2_unused_keys(self.h, up(keys))
        return [int(k) for k in keys[:n]]

    def factor_exists(self, i):
        return bool(self.L.orc_isam2_factor_exists(self.h, int(i)))

    def num_factors(self):
        """slots of getFactorsUnsafe(), removed ones included"""
        return self.L.orc_isam2_num_factors(self.h)

    def _values(self, which):
        n = self.L.orc_isam2_num_variables(self.h)
        keys = np.zeros(n, dtype=np.uint64)
        types = np.zeros(n, dtype=np.int32)
        assert self.L.orc_isam2_values(self.h, which, up(keys), ip(types), None) == 0
        packed = np.zeros(int(sum(VAR_STORE[t] for t in types)))
        assert self.L.orc_isam2_values(self.h, which, None, None, dp(packed)) == 0
        out, o = Values(), 0
        for k, t in zip(keys.tolist(), types.tolist()):
            out.insert(k, t, packed[o:o + VAR_STORE[t]])
            o += VAR_STORE[t]
        return out

    def calculateEstimate(self):
        return self._values(0)

    def calculateBestEstimate(self):
        return self._values(1)

    def getLinearizationPoint(self):
        return self._values(2)

    def getDelta(self):
        """{key: vector}"""
        lin = self._values(2)
        tot = sum(VAR_DIM[lin.type(k)] for k in lin.keys())
        d = np.zeros(tot)
        self.L.orc_isam2_delta(self.h, dp(d))
        out, o = {}, 0
        for k in lin.keys():
            n = VAR_DIM[lin.type(k)]
            out[k] = d[o:o + n].copy()
            o += n
        return out

    def cliques(self):
        """[(keys, n_frontal_keys, RSd (nf, n), parent index)] depth-first from the roots"""
        out = []
        for i in range(self.L.orc_isam2_snapshot(self.h)):
            info = np.zeros(5, dtype=np.int32)
            self.L.orc_isam2_clique_info(self.h, i, ip(info))
            keys = np.zeros(info[0], dtype=np.uint64)
            rsd = np.empty(info[2] * info[3])
            self.L.orc_isam2_clique_get(self.h, i, up(keys), dp(rsd))
            out.append(([int(k) for k in keys], int(info[1]), rsd.reshape(info[3], info[2]).T.copy(), int(info[4])))
        return out
