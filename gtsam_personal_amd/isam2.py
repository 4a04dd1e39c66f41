"""ISAM2 mirror over the C ABI (include/lmgpu.h, lmgpu_isam2_*): same names and argument meaning as the reference interface
  ISAM2Params / ISAM2GaussNewtonParams   gtsam/nonlinear/ISAM2Params.h:35-60, 133-246
  ISAM2::update / calculateEstimate / calculateBestEstimate / getLinearizationPoint / getDelta   gtsam/nonlinear/ISAM2.h:146-260
  ISAM2Result                             gtsam/nonlinear/ISAM2Result.h:60-93
All numerics run in liblmgpu.so on the GPU; this file only marshals arrays.  The fill-reducing ordering is the caller's: pass
`ccolamd`, a callable (n_rows, n_cols, col_ptr, row_idx, cmember) -> permutation that runs the CCOLAMD the reference side links
(Ordering::ColamdConstrained, gtsam/inference/Ordering.cpp:50-125, with its knobs)."""
from __future__ import annotations

import ctypes as ct

import numpy as np

from . import _lib
from .graph import FACTOR_ARITY, F_PRIOR_CAM, F_SFM, N_UNIT, VAR_DIM, VAR_STORE, VAR_STORE_DEV, NonlinearFactorGraph, Values


class ISAM2GaussNewtonParams:
    def __init__(self, wildfireThreshold=0.001):
        self.wildfireThreshold = wildfireThreshold


class ISAM2DoglegParams:
    """ISAM2DoglegParams(initialDelta, wildfireThreshold, adaptationMode) (ISAM2Params.h:68-110); adaptationMode as
    DoglegOptimizerImpl::TrustRegionAdaptationMode: 0 SEARCH_EACH_ITERATION, 1 SEARCH_REDUCE_ONLY, 2 ONE_STEP_PER_ITERATION"""
    SEARCH_EACH_ITERATION, SEARCH_REDUCE_ONLY, ONE_STEP_PER_ITERATION = 0, 1, 2

    def __init__(self, initialDelta=1.0, wildfireThreshold=1e-5, adaptationMode=0):
        self.initialDelta = initialDelta
        self.wildfireThreshold = wildfireThreshold
        self.adaptationMode = adaptationMode


class ISAM2Params:
    """ISAM2Params(optimizationParams, relinearizeThreshold, relinearizeSkip, enableRelinearization) — ISAM2Params.h:211-246.
    relinearizeThreshold: a double, or the FastMap<char, Vector> form as {symbol character: per-dof thresholds} (:139-141);
    enablePartialRelinearizationCheck (:214-222)"""

    def __init__(self, optimizationParams=None, relinearizeThreshold=0.1, relinearizeSkip=10, enableRelinearization=True,
                 enablePartialRelinearizationCheck=False, evaluateNonlinearError=False, findUnusedFactorSlots=False):
        self.optimizationParams = optimizationParams or ISAM2GaussNewtonParams()
        self.relinearizeThreshold = relinearizeThreshold
        self.relinearizeSkip = relinearizeSkip
        self.enableRelinearization = enableRelinearization
        self.enablePartialRelinearizationCheck = enablePartialRelinearizationCheck
        self.evaluateNonlinearError = evaluateNonlinearError  # ISAM2Params.h:200-203
        self.findUnusedFactorSlots = findUnusedFactorSlots  # ISAM2Params.h:225


class ISAM2Result:
    def __init__(self, r, errors=None):
        self.errorBefore, self.errorAfter = errors if errors else (None, None)  # with ISAM2Params.evaluateNonlinearError
        self.variablesRelinearized = r.variablesRelinearized
        self.variablesReeliminated = r.variablesReeliminated
        self.factorsRecalculated = r.factorsRecalculated
        self.cliques = r.cliques
        self.batch = bool(r.batch)

    def as_dict(self):
        return dict(variablesRelinearized=self.variablesRelinearized, variablesReeliminated=self.variablesReeliminated,
                    factorsRecalculated=self.factorsRecalculated, cliques=self.cliques, batch=int(self.batch))


class ISAM2:
    def __init__(self, params: ISAM2Params | None = None, ccolamd=None, device: int = 0):
        if ccolamd is None:
            raise ValueError("ISAM2 needs the caller's constrained COLAMD (ccolamd=...): the ordering is a boundary input")
        self.params = params or ISAM2Params()
        self.lib = _lib.load()
        self._ccolamd = ccolamd

        def _cb(user, n_rows, n_cols, col_ptr, row_idx, cmember, perm_out):
            try:
                cp = np.ctypeslib.as_array(col_ptr, shape=(n_cols + 1,))
                ri = np.ctypeslib.as_array(row_idx, shape=(max(1, int(cp[n_cols])),))
                cm = np.ctypeslib.as_array(cmember, shape=(n_cols,))
                np.ctypeslib.as_array(perm_out, shape=(n_cols,))[:] = self._ccolamd(n_rows, n_cols, cp, ri, cm)
                return 1
            except Exception:  # noqa: BLE001 -- must not unwind through the C caller
                import traceback
                traceback.print_exc()
                return 0

        self._cb = _lib.CCOLAMD_FN(_cb)  # keep alive
        p = self.params
        by_char = p.relinearizeThreshold if isinstance(p.relinearizeThreshold, dict) else None
        cp = _lib.lmgpu_isam2_params(0.1 if by_char is not None else float(p.relinearizeThreshold), int(p.relinearizeSkip),
                                     int(bool(p.enableRelinearization)), float(p.optimizationParams.wildfireThreshold))
        cfg = _lib.lmgpu_config(device, 0, 1, 0)
        self._h = ct.c_void_p()
        rc = self.lib.lmgpu_isam2_create(ct.byref(cfg), ct.byref(cp), ct.cast(self._cb, ct.c_void_p), None, ct.byref(self._h))
        self._check(rc)
        if by_char is not None:
            items = sorted(by_char.items())
            chrs = bytes(ord(c) if isinstance(c, str) else int(c) for c, _ in items)
            dims = np.asarray([len(v) for _, v in items], dtype=np.int32)
            vals = np.asarray([x for _, v in items for x in v], dtype=np.float64)
            self._check(self.lib.lmgpu_isam2_set_relinearize_thresholds(self._h, len(items), chrs, dims.ctypes.data_as(_lib._I),
                                                                        vals.ctypes.data_as(_lib._D)))
        if getattr(p, "enablePartialRelinearizationCheck", False):
            self._check(self.lib.lmgpu_isam2_set_partial_relinearization_check(self._h, 1))
        if getattr(p, "evaluateNonlinearError", False):
            self._check(self.lib.lmgpu_isam2_set_evaluate_nonlinear_error(self._h, 1))
        if getattr(p, "findUnusedFactorSlots", False):
            self._check(self.lib.lmgpu_isam2_set_find_unused_factor_slots(self._h, 1))
        if isinstance(p.optimizationParams, ISAM2DoglegParams):
            o = p.optimizationParams
            self._check(self.lib.lmgpu_isam2_set_dogleg(self._h, float(o.initialDelta), float(o.wildfireThreshold), int(o.adaptationMode)))
        self._u0v0 = {}  # constant principal points of Cal3Bundler cameras (do not travel, see lmgpu.h CAM_BUNDLER)

    def _check(self, rc):
        if rc == _lib.LMGPU_OK:
            return
        msg = self.lib.lmgpu_isam2_last_error(self._h) if self._h else b""
        if rc == _lib.LMGPU_INDETERMINATE:
            raise _lib.IndeterminantLinearSystemException(int(self.lib.lmgpu_isam2_last_failed_key(self._h)))
        raise _lib.LmgpuError(f"lmgpu status {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.lmgpu_isam2_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update(self, newFactors: NonlinearFactorGraph | None = None, newTheta: Values | None = None, removeFactorIndices=(),
               constrainedKeys=None, noRelinKeys=None, extraReelimKeys=None, force_relinearize=False, forceFullSolve=False) -> ISAM2Result:
        """ISAM2::update(newFactors, newTheta, removeFactorIndices, constrainedKeys, noRelinKeys, extraReelimKeys, force_relinearize)
        (gtsam/nonlinear/ISAM2.h:146-186, ISAM2UpdateParams.h:30-90).  removeFactorIndices: positions in getFactorsUnsafe();
        constrainedKeys: {key: group} (None = not given); the new factors of this update get the indices size() .. of the list (with
        ISAM2Params.findUnusedFactorSlots: its empty slots first)"""
        U64 = ct.POINTER(ct.c_uint64)
        if newTheta is not None and newTheta.size():
            keys = np.array(newTheta.keys(), dtype=np.uint64)
            types = np.array([newTheta.type(k) for k in newTheta.keys()], dtype=np.int32)
            packed = np.concatenate([newTheta.at(k)[:VAR_STORE_DEV[newTheta.type(k)]] for k in newTheta.keys()]).astype(np.float64)
            for k in newTheta.keys():
                if newTheta.type(k) == 3:
                    self._u0v0[k] = newTheta.at(k)[15:17].copy()
            self._check(self.lib.lmgpu_isam2_add_variables(self._h, len(keys), keys.ctypes.data_as(U64), types.ctypes.data_as(_lib._I),
                                                           packed.ctypes.data_as(_lib._D)))
        if newFactors is not None and newFactors.size():
            rec = [None] * newFactors.size()
            for ftype, kind, gi, keys, meas, noise, models in newFactors.buckets():
                for i, g in enumerate(gi.tolist()):
                    rec[g] = (ftype, kind, keys[i], meas[i], None if kind == N_UNIT else noise[i], models[i].robust_kind, models[i].robust_k)
            for ftype, kind, keys, meas, noise, rkind, rk in rec:  # one call per factor keeps the graph order across buckets
                m = np.array(meas, dtype=np.float64)
                if ftype == F_SFM:
                    m = m - self._u0v0[int(keys[0])]
                if ftype == F_PRIOR_CAM:
                    m = m[:15]
                kk = np.ascontiguousarray(keys[:FACTOR_ARITY[ftype]], dtype=np.uint64)
                nz = None if noise is None else np.ascontiguousarray(noise, dtype=np.float64)
                self._check(self.lib.lmgpu_isam2_add_factors_robust(self._h, ftype, 1, kk.ctypes.data_as(U64), np.ascontiguousarray(m).ctypes.data_as(_lib._D),
                                                                    kind, None if nz is None else nz.ctypes.data_as(_lib._D), int(rkind), float(rk)))
        res = _lib.lmgpu_isam2_result()
        rm = np.asarray(list(removeFactorIndices), dtype=np.uint64)
        ck = np.asarray(sorted(constrainedKeys) if constrainedKeys else [], dtype=np.uint64)
        cg = np.asarray([constrainedKeys[int(k)] for k in ck], dtype=np.int32)
        nr = np.asarray(list(noRelinKeys or []), dtype=np.uint64)
        ex = np.asarray(list(extraReelimKeys or []), dtype=np.uint64)
        up = _lib.lmgpu_isam2_update_params(len(rm), rm.ctypes.data_as(U64), int(constrainedKeys is not None), len(ck), ck.ctypes.data_as(U64),
                                            cg.ctypes.data_as(_lib._I), len(nr), nr.ctypes.data_as(U64), len(ex), ex.ctypes.data_as(U64),
                                            int(force_relinearize), int(forceFullSolve))
        self._check(self.lib.lmgpu_isam2_update_with(self._h, ct.byref(up), ct.byref(res)))
        errors = None
        if getattr(self.params, "evaluateNonlinearError", False):
            b, a = ct.c_double(), ct.c_double()
            self._check(self.lib.lmgpu_isam2_get_errors(self._h, ct.byref(b), ct.byref(a)))
            errors = (b.value, a.value)
        return ISAM2Result(res, errors)

    def error(self, which=0):
        """getFactorsUnsafe().error(calculateEstimate()) (which = 0) or .error(getLinearizationPoint()) (which = 2)"""
        e = ct.c_double()
        self._check(self.lib.lmgpu_isam2_error(self._h, int(which), ct.byref(e)))
        return e.value

    def marginalCovariance(self, key):
        """ISAM2::marginalCovariance(key) (gtsam/nonlinear/ISAM2.h:253-257)"""
        lin_type = None
        n = self.lib.lmgpu_isam2_num_variables(self._h)
        keys = np.zeros(n, dtype=np.uint64)
        types = np.zeros(n, dtype=np.int32)
        self._check(self.lib.lmgpu_isam2_get_values(self._h, 2, keys.ctypes.data_as(ct.POINTER(ct.c_uint64)), types.ctypes.data_as(_lib._I), None))
        at = np.nonzero(keys == np.uint64(key))[0]
        if at.size:
            lin_type = int(types[at[0]])
        d = VAR_DIM[lin_type] if lin_type is not None else 9
        cov = np.zeros((d, d))
        self._check(self.lib.lmgpu_isam2_marginal_covariance(self._h, int(key), cov.ctypes.data_as(_lib._D)))
        return cov

    def doglegDelta(self):
        """the current trust-region radius (ISAM2::doglegDelta_) with ISAM2DoglegParams"""
        return float(self.lib.lmgpu_isam2_get_dogleg_delta(self._h))

    def unusedKeys(self):
        """ISAM2Result::unusedKeys of the last update: the variables that left the system with their last factor"""
        n = self.lib.lmgpu_isam2_get_unused_keys(self._h, None)
        keys = np.zeros(max(n, 1), dtype=np.uint64)
        self.lib.lmgpu_isam2_get_unused_keys(self._h, keys.ctypes.data_as(ct.POINTER(ct.c_uint64)))
        return [int(k) for k in keys[:n]]

    def marginalizeLeaves(self, leafKeys):
        """ISAM2::marginalizeLeaves(leafKeys, &marginalFactorsIndices, &deletedFactorsIndices) (gtsam/nonlinear/ISAM2.h:198-222); returns
        (marginalFactorsIndices, deletedFactorsIndices).  A key that is not a leaf is refused before anything changes (LmgpuError)."""
        keys = np.asarray([int(k) for k in leafKeys], dtype=np.uint64)
        U = ct.POINTER(ct.c_uint64)
        nm, nd = ct.c_int32(0), ct.c_int32(0)
        self._check(self.lib.lmgpu_isam2_marginalize_leaves(self._h, len(keys), keys.ctypes.data_as(U), ct.byref(nm), ct.byref(nd)))
        mi, di = np.zeros(max(nm.value, 1), dtype=np.uint64), np.zeros(max(nd.value, 1), dtype=np.uint64)
        self._check(self.lib.lmgpu_isam2_get_marginalize_result(self._h, mi.ctypes.data_as(U), di.ctypes.data_as(U)))
        return [int(i) for i in mi[:nm.value]], [int(i) for i in di[:nd.value]]

    def getFixedVariables(self):
        """ISAM2::getFixedVariables() (ISAM2.h:259): the keys of the marginal factors, ascending"""
        n = self.lib.lmgpu_isam2_get_fixed_variables(self._h, None)
        keys = np.zeros(max(n, 1), dtype=np.uint64)
        self.lib.lmgpu_isam2_get_fixed_variables(self._h, keys.ctypes.data_as(ct.POINTER(ct.c_uint64)))
        return [int(k) for k in keys[:n]]

    def marginal_factor(self, i):
        """parity tap: slot i of getFactorsUnsafe() as the LinearContainerFactor marginalizeLeaves left there -> (keys, dims, augmented
        information matrix), None when the slot holds none"""
        n = self.lib.lmgpu_isam2_get_marginal_factor(self._h, int(i), None, None, None)
        if n < 0:
            return None
        keys, dims = np.zeros(max(n, 1), dtype=np.uint64), np.zeros(max(n, 1), dtype=np.int32)
        self.lib.lmgpu_isam2_get_marginal_factor(self._h, int(i), keys.ctypes.data_as(ct.POINTER(ct.c_uint64)), dims.ctypes.data_as(_lib._I), None)
        N = int(dims[:n].sum()) + 1
        info = np.zeros(N * N)
        if self.lib.lmgpu_isam2_get_marginal_factor(self._h, int(i), None, None, info.ctypes.data_as(_lib._D)) < 0:
            raise _lib.LmgpuError("reading a marginal factor failed")
        return [int(k) for k in keys[:n]], [int(d) for d in dims[:n]], info.reshape(N, N).T.copy()

    def factor_exists(self, i):
        """getFactorsUnsafe().exists(i)"""
        return bool(self.lib.lmgpu_isam2_factor_exists(self._h, int(i)))

    def num_factors(self):
        """getFactorsUnsafe().size(): slots, removed ones included"""
        return self.lib.lmgpu_isam2_num_factors(self._h)

    def _values(self, which) -> Values:
        n = self.lib.lmgpu_isam2_num_variables(self._h)
        keys = np.zeros(n, dtype=np.uint64)
        types = np.zeros(n, dtype=np.int32)
        self._check(self.lib.lmgpu_isam2_get_values(self._h, 2, keys.ctypes.data_as(ct.POINTER(ct.c_uint64)), types.ctypes.data_as(_lib._I), None))
        packed = np.zeros(int(sum(VAR_STORE_DEV[t] for t in types)))
        self._check(self.lib.lmgpu_isam2_get_values(self._h, which, None, None, packed.ctypes.data_as(_lib._D)))
        out, o = Values(), 0
        for k, t in zip(keys.tolist(), types.tolist()):
            v = np.zeros(VAR_STORE[t])
            v[:VAR_STORE_DEV[t]] = packed[o:o + VAR_STORE_DEV[t]]
            if t == 3:
                v[15:17] = self._u0v0[k]
            out.insert(k, t, v)
            o += VAR_STORE_DEV[t]
        return out

    def calculateEstimate(self, key=None):
        """ISAM2::calculateEstimate() -> Values, or calculateEstimate(key) -> the packed value of one variable (ISAM2.cpp:748-760)"""
        if key is None:
            return self._values(0)
        return self._value(0, key)

    def _value(self, which, key):
        t = ct.c_int32()
        buf = np.zeros(17)
        self._check(self.lib.lmgpu_isam2_get_value(self._h, which, int(key), ct.byref(t), buf.ctypes.data_as(_lib._D)))
        out = buf[:VAR_STORE[t.value]].copy()
        if t.value == 3:  # the constant principal point of a Cal3Bundler camera does not travel
            out[15:17] = self._u0v0[int(key)]
        return out

    def calculateBestEstimate(self) -> Values:
        return self._values(1)

    def getLinearizationPoint(self) -> Values:
        return self._values(2)

    def getDelta(self):
        """{key: vector} (VectorValues)"""
        lin = self._values(2)
        d = np.zeros(sum(VAR_DIM[lin.type(k)] for k in lin.keys()))
        self._check(self.lib.lmgpu_isam2_get_delta(self._h, d.ctypes.data_as(_lib._D)))
        out, o = {}, 0
        for k in lin.keys():
            n = VAR_DIM[lin.type(k)]
            out[k] = d[o:o + n].copy()
            o += n
        return out

    def size(self):
        return self.lib.lmgpu_isam2_num_variables(self._h)

    def cliques(self):
        """parity tap: [(keys, n_frontal_keys, RSd (nf, n), parent index)] depth-first from the roots"""
        out = []
        for i in range(self.lib.lmgpu_isam2_num_cliques(self._h)):
            info = np.zeros(5, dtype=np.int32)
            self._check(self.lib.lmgpu_isam2_clique_info(self._h, i, info.ctypes.data_as(_lib._I)))
            keys = np.zeros(info[0], dtype=np.uint64)
            rsd = np.empty(info[2] * info[3])
            self._check(self.lib.lmgpu_isam2_get_clique(self._h, i, keys.ctypes.data_as(ct.POINTER(ct.c_uint64)), rsd.ctypes.data_as(_lib._D)))
            out.append(([int(k) for k in keys], int(info[1]), rsd.reshape(info[3], info[2]).T.copy(), int(info[4])))
        return out
