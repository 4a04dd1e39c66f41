"""Host-side mirror of the reference's graph-building interface for the LM hot path.

Names follow the reference (citations relative to the reference tree):
  Symbol / symbol_shorthand      gtsam/inference/Symbol.h, Symbol.cpp:29-46  (char << 56 | index)
  Values                         gtsam/nonlinear/Values.h (ordered by Key)
  noiseModel.*                   gtsam/linear/NoiseModel.cpp ("smart" constructors :83-112, :280-312)
  NonlinearFactorGraph           gtsam/nonlinear/NonlinearFactorGraph.h
  GeneralSFMFactor, BetweenFactor, PriorFactor, GenericProjectionFactor
                                 gtsam/slam/GeneralSFMFactor.h, BetweenFactor.h, nonlinear/PriorFactor.h, slam/ProjectionFactor.h
Everything here is plain numpy bookkeeping: the numbers are evaluated on the GPU through the C ABI
(include/lmgpu.h).  Factors are kept in bulk arrays per (factor type, noise kind) bucket.
"""
from __future__ import annotations

import numpy as np

# ---- enums shared with include/lmgpu.h
POSE2, POSE3, POINT3, CAM_BUNDLER, POINT2, CAL3_S2 = 0, 1, 2, 3, 4, 5
VAR_DIM = (3, 6, 3, 9, 2, 5)
VAR_STORE = (3, 12, 3, 17, 2, 5)      # host packed value (camera keeps u0, v0)
VAR_STORE_DEV = (3, 12, 3, 15, 2, 5)  # C-ABI packed value

(F_SFM, F_BETWEEN_POSE2, F_BETWEEN_POSE3, F_PRIOR_POSE2, F_PRIOR_POSE3, F_PRIOR_POINT3, F_PRIOR_CAM, F_PROJECTION, F_PROJECTION_BPS,
 F_BEARING_RANGE_2D, F_SFM2, F_PRIOR_CAL3_S2) = range(12)
FACTOR_ARITY = (2, 2, 2, 1, 1, 1, 1, 2, 2, 2, 3, 1)
FACTOR_ROWS = (2, 3, 6, 3, 6, 3, 9, 2, 2, 2, 2, 5)
FACTOR_MEAS = (2, 3, 12, 3, 12, 3, 17, 7, 19, 2, 2, 5)  # host measurement doubles (PRIOR_CAM carries u0, v0)
FACTOR_VARS = ((CAM_BUNDLER, POINT3), (POSE2, POSE2), (POSE3, POSE3), (POSE2,), (POSE3,), (POINT3,), (CAM_BUNDLER,), (POSE3, POINT3), (POSE3, POINT3),
               (POSE2, POINT2), (POSE3, POINT3, CAL3_S2), (CAL3_S2,))

N_UNIT, N_ISO, N_DIAG, N_GAUSS = 0, 1, 2, 3


def symbol(c: str, j: int) -> int:
    """Symbol(c, j).key()  (gtsam/inference/Symbol.cpp:40-48)."""
    return (ord(c) << 56) | int(j)


def C(j):  # symbol_shorthand::C
    return symbol("c", j)


def P(j):
    return symbol("p", j)


def X(j):
    return symbol("x", j)


def L(j):
    return symbol("l", j)


# ---------------------------------------------------------------- noise models
class NoiseModel:
    """kind: N_UNIT | N_ISO (sigma) | N_DIAG (sigmas) | N_GAUSS (sqrt information R, row-major)."""

    def __init__(self, dim, kind, data=None):
        self.dim, self.kind = int(dim), kind
        self.data = None if data is None else np.asarray(data, dtype=np.float64)
        self.robust_kind, self.robust_k = 0, 0.0  # noiseModel.Robust: m-estimator id (lmgpu_robust_kind) and constant

    def invsigmas(self):
        if self.kind == N_UNIT:
            return np.ones(self.dim)
        if self.kind == N_ISO:
            return np.full(self.dim, 1.0 / float(self.data))
        if self.kind == N_DIAG:
            return 1.0 / self.data
        raise ValueError("gaussian model has no sigmas")

    def __repr__(self):
        return f"NoiseModel(dim={self.dim}, kind={self.kind})"


class noiseModel:
    class Unit:
        @staticmethod
        def Create(dim):
            return NoiseModel(dim, N_UNIT)

    class Isotropic:
        @staticmethod
        def Sigma(dim, sigma, smart=True):
            # NoiseModel.cpp:600-607
            if smart and abs(sigma - 1.0) < 1e-9:
                return NoiseModel(dim, N_UNIT)
            return NoiseModel(dim, N_ISO, float(sigma))

        @staticmethod
        def Variance(dim, variance, smart=True):
            if smart and abs(variance - 1.0) < 1e-9:
                return NoiseModel(dim, N_UNIT)
            return NoiseModel(dim, N_ISO, float(np.sqrt(variance)))

    class Diagonal:
        @staticmethod
        def Sigmas(sigmas, smart=True):
            sigmas = np.asarray(sigmas, dtype=np.float64)
            if smart and sigmas.size:
                if (sigmas < 1e-8).any():
                    raise NotImplementedError("constrained noise models take the QR path (out of scope, SURVEY section 2)")
                if (sigmas == sigmas[0]).all():
                    return noiseModel.Isotropic.Sigma(sigmas.size, sigmas[0], True)
            return NoiseModel(sigmas.size, N_DIAG, sigmas)

        @staticmethod
        def Variances(variances, smart=True):
            variances = np.asarray(variances, dtype=np.float64)
            if smart and (variances == variances[0]).all():
                return noiseModel.Isotropic.Variance(variances.size, variances[0], True)
            return NoiseModel(variances.size, N_DIAG, np.sqrt(variances))

        @staticmethod
        def Precisions(precisions, smart=True):
            return noiseModel.Diagonal.Variances(1.0 / np.asarray(precisions, dtype=np.float64), smart)

    class Gaussian:
        @staticmethod
        def Information(info, smart=True):
            """NoiseModel.cpp:98-112: diagonal -> Diagonal::Precisions, else R = chol(info) upper."""
            info = np.asarray(info, dtype=np.float64)
            n = info.shape[0]
            if smart and np.count_nonzero(info - np.diag(np.diag(info))) == 0:
                return noiseModel.Diagonal.Precisions(np.diag(info).copy(), True)
            R = np.linalg.cholesky(info).T  # upper, info = R^T R
            return NoiseModel(n, N_GAUSS, R.copy())

        @staticmethod
        def SqrtInformation(R, smart=True):
            R = np.asarray(R, dtype=np.float64)
            if smart and np.count_nonzero(R - np.diag(np.diag(R))) == 0:
                return noiseModel.Diagonal.Sigmas(1.0 / np.diag(R), True)
            return NoiseModel(R.shape[0], N_GAUSS, R.copy())

        @staticmethod
        def Covariance(cov, smart=True):
            cov = np.asarray(cov, dtype=np.float64)
            if smart and np.count_nonzero(cov - np.diag(np.diag(cov))) == 0:
                return noiseModel.Diagonal.Variances(np.diag(cov).copy(), True)
            return noiseModel.Gaussian.Information(np.linalg.inv(cov), False)


class _MEstimator:
    def __init__(self, kind, k):
        self.kind, self.k = int(kind), float(k)


def _mest(kind, default_k):
    class _E:
        @staticmethod
        def Create(k=default_k):
            return _MEstimator(kind, k)
    return _E


class mEstimator:
    """gtsam/linear/LossFunctions.h m-estimators (default constants as there); ids = lmgpu_robust_kind."""
    Fair = _mest(1, 1.3998)
    Huber = _mest(2, 1.345)
    Cauchy = _mest(3, 0.1)
    Tukey = _mest(4, 4.6851)
    Welsch = _mest(5, 2.9846)
    GemanMcClure = _mest(6, 1.0)
    DCS = _mest(7, 1.0)
    L2WithDeadZone = _mest(8, 1.0)


class _Robust:
    @staticmethod
    def Create(robust: _MEstimator, noise: NoiseModel):
        """noiseModel::Robust::Create(mEstimator, Gaussian model) (gtsam/linear/NoiseModel.cpp:732-735)"""
        m = NoiseModel(noise.dim, noise.kind, noise.data)
        m.robust_kind, m.robust_k = robust.kind, robust.k
        return m


noiseModel.Robust = _Robust
noiseModel.mEstimator = mEstimator


# ---------------------------------------------------------------- values
def pose3_pack(R, t):
    return np.concatenate([np.asarray(R, dtype=np.float64).reshape(9), np.asarray(t, dtype=np.float64).reshape(3)])


def camera_pack(R, t, f, k1, k2, u0=0.0, v0=0.0):
    return np.concatenate([pose3_pack(R, t), [f, k1, k2, u0, v0]])


class Values:
    """key -> (type, packed value).  Iteration order is ascending Key like the reference's std::map."""

    def __init__(self):
        self._type = {}
        self._val = {}

    def insert(self, key, vtype, value):
        key = int(key)
        if key in self._type:
            raise KeyError(f"key {key} already in Values")
        value = np.asarray(value, dtype=np.float64).reshape(-1)
        if value.size != VAR_STORE[vtype]:
            raise ValueError(f"value for type {vtype} needs {VAR_STORE[vtype]} doubles")
        self._type[key] = vtype
        self._val[key] = value.copy()

    def insert_pose2(self, key, x, y, theta):
        self.insert(key, POSE2, [x, y, theta])

    def insert_pose3(self, key, R, t):
        self.insert(key, POSE3, pose3_pack(R, t))

    def insert_point3(self, key, p):
        self.insert(key, POINT3, p)

    def insert_point2(self, key, p):
        self.insert(key, POINT2, p)

    def insert_cal3_s2(self, key, fx, fy, s, u0, v0):
        """Cal3_S2 as a variable (self-calibration, examples/SelfCalibrationExample.cpp:55-56)"""
        self.insert(key, CAL3_S2, [fx, fy, s, u0, v0])

    def insert_camera(self, key, R, t, f, k1, k2, u0=0.0, v0=0.0):
        self.insert(key, CAM_BUNDLER, camera_pack(R, t, f, k1, k2, u0, v0))

    def keys(self):
        return sorted(self._type)

    def type(self, key):
        return self._type[int(key)]

    def at(self, key):
        return self._val[int(key)]

    def update(self, key, value):
        self._val[int(key)] = np.asarray(value, dtype=np.float64).reshape(-1).copy()

    def exists(self, key):
        return int(key) in self._type

    def erase(self, key):
        """Values::erase (gtsam/nonlinear/Values.cpp)"""
        key = int(key)
        if key not in self._type:
            raise KeyError(f"key {key} not in Values")
        del self._type[key]
        del self._val[key]

    def size(self):
        return len(self._type)

    def dim(self):
        return sum(VAR_DIM[t] for t in self._type.values())

    def copy(self):
        v = Values()
        v._type = dict(self._type)
        v._val = {k: a.copy() for k, a in self._val.items()}
        return v


# ---------------------------------------------------------------- factors
class _Bucket:
    def __init__(self, ftype, noise_kind):
        self.ftype, self.noise_kind = ftype, noise_kind
        self.graph_index, self.keys, self.meas, self.noise = [], [], [], []
        self.models = []  # original NoiseModel per factor (for host-side consumers)

    def finalize(self):
        ar, ml = FACTOR_ARITY[self.ftype], FACTOR_MEAS[self.ftype]
        gi = np.concatenate([np.atleast_1d(np.asarray(g, dtype=np.int64)) for g in self.graph_index])
        keys = np.concatenate([np.asarray(k, dtype=np.uint64).reshape(-1, ar) for k in self.keys])
        meas = np.concatenate([np.asarray(m, dtype=np.float64).reshape(-1, ml) for m in self.meas])
        noise = None
        if self.noise_kind != N_UNIT:
            noise = np.concatenate([np.asarray(n, dtype=np.float64).reshape(meas_n, -1) for n, meas_n in self.noise])
        return gi, keys, meas, noise


class NonlinearFactorGraph:
    """Factors are appended in graph order; each gets the next graph index (its position in the reference's
    FactorGraph vector).  Storage is bucketed by (factor type, device noise kind)."""

    def __init__(self):
        self._buckets = {}
        self._n = 0
        self._final = None

    def size(self):
        return self._n

    def _add(self, ftype, keys, meas, noise: NoiseModel):
        """keys: (n, arity) ; meas: (n, FACTOR_MEAS) ; one noise model shared by the n factors."""
        keys = np.asarray(keys, dtype=np.uint64).reshape(-1, FACTOR_ARITY[ftype])
        n = keys.shape[0]
        meas = np.asarray(meas, dtype=np.float64).reshape(n, FACTOR_MEAS[ftype])
        rows = FACTOR_ROWS[ftype]
        if noise is None:
            noise = NoiseModel(rows, N_UNIT)
        if noise.dim != rows:
            raise ValueError(f"noise model dim {noise.dim} != factor rows {rows}")
        if noise.kind == N_UNIT:
            dev_kind, ndata = N_UNIT, None
        elif noise.kind in (N_ISO, N_DIAG):
            dev_kind, ndata = N_DIAG, np.tile(noise.invsigmas(), (n, 1))
        else:
            dev_kind, ndata = N_GAUSS, np.tile(noise.data.reshape(1, rows * rows), (n, 1))
        b = self._buckets.setdefault((ftype, dev_kind, noise.robust_kind, noise.robust_k), _Bucket(ftype, dev_kind))
        b.graph_index.append(np.arange(self._n, self._n + n, dtype=np.int64))
        b.keys.append(keys)
        b.meas.append(meas)
        if dev_kind != N_UNIT:
            b.noise.append((ndata, n))
        b.models.extend([noise] * n)
        self._n += n
        self._final = None

    def push_back(self, other: "NonlinearFactorGraph"):
        """FactorGraph::push_back(const FactorGraph&) (gtsam/inference/FactorGraph.h): append the factors of `other` in its graph order"""
        rec = [None] * other.size()
        for ftype, _, gi, keys, meas, _, models in other.buckets():
            for i, g in enumerate(gi.tolist()):
                rec[g] = (ftype, keys[i], meas[i], models[i])
        for ftype, keys, meas, model in rec:
            self._add(ftype, [keys], [meas], model)

    def resize(self, n):
        """FactorGraph::resize(0) as the incremental examples use it to start the next batch of new factors"""
        if n != 0:
            raise NotImplementedError("only resize(0)")
        self.__init__()

    # -- reference-named adders (single factor or a batch with one shared model)
    def add_GeneralSFMFactor(self, measured, model, cameraKey, landmarkKey):
        """GeneralSFMFactor<PinholeCamera<Cal3Bundler>, Point3>(measured, model, cameraKey, landmarkKey)"""
        keys = np.stack([np.asarray(cameraKey, dtype=np.uint64).reshape(-1), np.asarray(landmarkKey, dtype=np.uint64).reshape(-1)], axis=1)
        self._add(F_SFM, keys, measured, model)

    def add_BetweenFactorPose2(self, key1, key2, measured, model):
        self._add(F_BETWEEN_POSE2, [[key1, key2]], measured, model)

    def add_BetweenFactorPose3(self, key1, key2, measured_R, measured_t, model):
        self._add(F_BETWEEN_POSE3, [[key1, key2]], pose3_pack(measured_R, measured_t), model)

    def add_BearingRangeFactor2D(self, poseKey, pointKey, measuredBearing, measuredRange, model):
        """BearingRangeFactor<Pose2, Point2>(poseKey, pointKey, Rot2::fromAngle(measuredBearing), measuredRange, model)
        (gtsam/sam/BearingRangeFactor.h:51-55); bearing in radians"""
        self._add(F_BEARING_RANGE_2D, [[poseKey, pointKey]], [measuredBearing, measuredRange], model)

    def add_PriorFactorPose2(self, key, prior, model):
        self._add(F_PRIOR_POSE2, [[key]], prior, model)

    def add_PriorFactorPose3(self, key, R, t, model):
        self._add(F_PRIOR_POSE3, [[key]], pose3_pack(R, t), model)

    def add_PriorFactorPoint3(self, key, p, model):
        self._add(F_PRIOR_POINT3, [[key]], p, model)

    def add_PriorFactorCamera(self, key, packed17, model):
        self._add(F_PRIOR_CAM, [[key]], packed17, model)

    def add_PriorFactorCal3_S2(self, key, K, model):
        """PriorFactor<Cal3_S2>; K = (fx, fy, s, u0, v0)"""
        self._add(F_PRIOR_CAL3_S2, [[key]], K, model)

    def add_GeneralSFMFactor2(self, measured, model, poseKey, landmarkKey, calibKey):
        """GeneralSFMFactor2<Cal3_S2>(measured, model, poseKey, landmarkKey, calibKey)  (gtsam/slam/GeneralSFMFactor.h:236-237)"""
        self._add(F_SFM2, [[poseKey, landmarkKey, calibKey]], measured, model)

    def add_GenericProjectionFactor(self, measured, model, poseKey, pointKey, K, body_P_sensor=None):
        """GenericProjectionFactor<Pose3, Point3, Cal3_S2>; K = (fx, fy, s, u0, v0); body_P_sensor = (R 3x3, t 3) or None
        (gtsam/slam/ProjectionFactor.h:99-113)"""
        zk = np.concatenate([np.asarray(measured, dtype=np.float64), np.asarray(K, dtype=np.float64)])
        if body_P_sensor is None:
            self._add(F_PROJECTION, [[poseKey, pointKey]], zk, model)
        else:
            self._add(F_PROJECTION_BPS, [[poseKey, pointKey]], np.concatenate([zk, pose3_pack(body_P_sensor[0], body_P_sensor[1])]), model)

    # -- bulk access
    def buckets(self):
        """[(ftype, dev_noise_kind, graph_index[n], keys[n, arity], meas[n, ml], noise[n, nl] | None, models[n])]"""
        if self._final is None:
            out = []
            for (ftype, kind, _rk, _rc), b in self._buckets.items():
                gi, keys, meas, noise = b.finalize()
                out.append((ftype, kind, gi, keys, meas, noise, b.models))
            self._final = out
        return self._final

    def keys(self):
        ks = set()
        for ftype, _, _, keys, _, _, _ in self.buckets():
            ks.update(np.unique(keys).tolist())
        return sorted(ks)

    def factor_keys_in_graph_order(self):
        """list over graph index of tuples of keys (VariableIndex input)."""
        out = [None] * self._n
        for ftype, _, gi, keys, _, _, _ in self.buckets():
            for g, k in zip(gi.tolist(), keys.tolist()):
                out[g] = tuple(int(x) for x in k)
        return out


# ---------------------------------------------------------------- orderings (boundary INPUT of the hot path)
class Ordering(list):
    """List of keys in elimination order (gtsam/inference/Ordering.h).  COLAMD / METIS orderings are computed by
    the reference's host code at optimizer construction (LevenbergMarquardtParams.h:112-117) and handed over;
    the helpers here cover the orderings that need no third-party code."""

    @staticmethod
    def Natural(graph: NonlinearFactorGraph):
        return Ordering(graph.keys())

    @staticmethod
    def Schur(graph: NonlinearFactorGraph, values: Values):
        """points (landmarks) first, then everything else, each by ascending key (timing/timeSFMBAL.h:64-96 style)."""
        ks = graph.keys()
        pts = [k for k in ks if values.type(k) == POINT3]
        rest = [k for k in ks if values.type(k) != POINT3]
        return Ordering(pts + rest)
