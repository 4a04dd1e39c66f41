"""Wire formats either side of the hot path: BAL, g2o / TORO 2D, EDGE3 / g2o 3D.

Restates the readers the reference's examples use so that the harness ingests the same numbers:
  SfmData::FromBalFile           gtsam/sfm/SfmData.cpp:189-246  (parses through FLOAT, negates v, openGL2gtsam :79-85)
  load2D / readG2o               gtsam/slam/dataset.cpp:505-569, 621-633 (noise: createNoiseModel :216-296)
  load3D                         gtsam/slam/dataset.cpp:922-944 (EDGE3 / EDGE_SE3:QUAT :811-863, VERTEX3 / VERTEX_SE3:QUAT :758-776)
"""
from __future__ import annotations

import numpy as np

from .graph import L, NonlinearFactorGraph, P, Values, noiseModel, symbol


# ---------------------------------------------------------------- small host geometry (loader-side only)
def rot3_expmap(w):
    """Rot3::Rodrigues = SO3::Expmap (gtsam/geometry/SO3.cpp:61-95)."""
    w = np.asarray(w, dtype=np.float64)
    theta2 = float(w @ w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if theta2 <= np.finfo(np.float64).eps:
        A, B = 1.0 - theta2 / 6.0, 0.5 - theta2 / 24.0
    else:
        theta = np.sqrt(theta2)
        A = np.sin(theta) / theta
        s2 = np.sin(theta / 2.0)
        B = 2.0 * s2 * s2 / theta2
    return np.eye(3) + A * W + B * (W @ W)


def rot3_rzryrx(x, y, z):
    """Rot3::RzRyRx (gtsam/geometry/Rot3M.cpp:84-108)."""
    cx, sx, cy, sy, cz, sz = np.cos(x), np.sin(x), np.cos(y), np.sin(y), np.cos(z), np.sin(z)
    ss_, cs_, sc_, cc_ = sx * sy, cx * sy, sx * cy, cx * cy
    c_s, s_s, _cs, _cc, s_c, c_c = cx * sz, sx * sz, cy * sz, cy * cz, sx * cz, cx * cz
    ssc, csc, sss, css = ss_ * cz, cs_ * cz, ss_ * sz, cs_ * sz
    return np.array([[_cc, -c_s + ssc, s_s + csc], [_cs, c_c + sss, -s_c + css], [-sy, sc_, cc_]])


def rot3_ypr(y, p, r):
    return rot3_rzryrx(r, p, y)


def rot3_quat(w, x, y, z):
    """Eigen::Quaternion::toRotationMatrix (normalised by the parser, dataset.cpp:737-743)."""
    nrm = np.sqrt(w * w + x * x + y * y + z * z)
    f = 1.0 / nrm
    w, x, y, z = f * w, f * x, f * y, f * z
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def pose3_compose(Ra, ta, Rb, tb):
    return Ra @ Rb, ta + Ra @ tb


# ---------------------------------------------------------------- BAL
class SfmData:
    def __init__(self):
        self.cameras = []  # (R 3x3, t 3, f, k1, k2)
        self.tracks = []   # dict(p=3, measurements=[(i, (u, v))])

    def numberCameras(self):
        return len(self.cameras)

    def numberTracks(self):
        return len(self.tracks)

    @staticmethod
    def FromBalFile(filename):
        """every number goes through float32 like the reference's `float u, v; is >> ...`."""
        with open(filename) as fh:
            tok = fh.read().split()
        it = iter(tok)
        nP, nT, nO = int(next(it)), int(next(it)), int(next(it))
        d = SfmData()
        d.tracks = [dict(p=None, measurements=[]) for _ in range(nT)]
        f32 = lambda s: float(np.float32(s))  # noqa: E731
        for _ in range(nO):
            i, j = int(next(it)), int(next(it))
            u, v = f32(next(it)), f32(next(it))
            d.tracks[j]["measurements"].append((i, (u, -v)))
        R90 = np.diag([1.0, -1.0, -1.0])  # openGLFixedRotation
        for _ in range(nP):
            w = [f32(next(it)) for _ in range(3)]
            t = np.array([f32(next(it)) for _ in range(3)])
            R = rot3_expmap(w)
            wRc = R.T @ R90                      # openGL2gtsam
            wtc = R.T @ (-t)
            f, k1, k2 = f32(next(it)), f32(next(it)), f32(next(it))
            d.cameras.append((wRc, wtc, f, k1, k2))
        for j in range(nT):
            d.tracks[j]["p"] = np.array([f32(next(it)) for _ in range(3)])
        return d


def bal_graph(db: SfmData, noise=None, camera_key=lambda i: i, point_key=P):
    """graph of tests/testGeneralSFMFactorB.cpp:44-63 (cameras keyed by bare index, points P(j)) by default;
    pass camera_key=C for examples/SFMExample_bal.cpp."""
    graph = NonlinearFactorGraph()
    noise = noise if noise is not None else noiseModel.Unit.Create(2)
    cams, pts, zs = [], [], []
    for j, tr in enumerate(db.tracks):
        for i, uv in tr["measurements"]:
            cams.append(camera_key(i))
            pts.append(point_key(j))
            zs.append(uv)
    graph.add_GeneralSFMFactor(np.array(zs), noise, np.array(cams, dtype=np.uint64), np.array(pts, dtype=np.uint64))
    initial = Values()
    for i, (R, t, f, k1, k2) in enumerate(db.cameras):
        initial.insert_camera(camera_key(i), R, t, f, k1, k2)
    for j, tr in enumerate(db.tracks):
        initial.insert_point3(point_key(j), tr["p"])
    return graph, initial


# ---------------------------------------------------------------- 2D: g2o / TORO
def _noise2d(v, smart, fmt):
    """createNoiseModel, gtsam/slam/dataset.cpp:216-300: the six numbers of an edge line as information (g2o, toro) or covariance
    (graph, cov) matrix in the G2O/COV or TORO/GRAPH ordering; "auto" guesses GRAPH vs COV from the zero pattern (:219-232)."""
    v = [float(x) for x in v]
    if fmt == "auto":
        if v[0] != 0.0 and v[1] == 0.0 and v[2] != 0.0 and v[3] != 0.0 and v[4] == 0.0 and v[5] == 0.0:
            fmt = "graph"
        elif v[0] != 0.0 and v[1] == 0.0 and v[2] == 0.0 and v[3] != 0.0 and v[4] == 0.0 and v[5] != 0.0:
            fmt = "cov"
        else:
            raise ValueError("load2D: unrecognized covariance matrix format in dataset file. Please specify the noise format.")
    if fmt in ("g2o", "cov"):
        if v[0] == 0.0 or v[3] == 0.0 or v[5] == 0.0:
            raise RuntimeError("load2D::readNoiseModel looks like this is not G2O matrix order")
        M = np.array([[v[0], v[1], v[2]], [v[1], v[3], v[4]], [v[2], v[4], v[5]]])
    elif fmt in ("toro", "graph"):
        if v[0] == 0.0 or v[2] == 0.0 or v[3] == 0.0:
            raise ValueError("load2D::readNoiseModel looks like this is not TORO matrix order")
        M = np.array([[v[0], v[1], v[4]], [v[1], v[2], v[5]], [v[4], v[5], v[3]]])
    else:
        raise ValueError(fmt)
    if fmt in ("g2o", "toro"):
        return noiseModel.Gaussian.Information(M, smart)
    return noiseModel.Gaussian.Covariance(M, smart)


def load2D(filename, noise_format="auto", smart=True, max_index=0):
    """(graph, initial): Pose2 vertices keyed by their integer id, Point2 landmarks by L(id) (gtsam/slam/dataset.cpp:505-569;
    noise_format as the reference's NoiseFormat: "auto" (default), "g2o", "toro", "graph", "cov"; readG2o passes "g2o").
    Tags: VERTEX2 / VERTEX_SE2 / VERTEX, VERTEX_XY (:192-205), EDGE2 / EDGE / EDGE_SE2 / ODOMETRY (:338-377), and the
    bearing-range measurements BR and LANDMARK (:450-502: a LANDMARK line's (x, y) sighting becomes bearing = atan2(y, x),
    range = |(x, y)| with sigmas (sqrt(v1 / 10), sqrt(v1)) when its covariance is isotropic, else (1, 1)).  max_index as the
    reference's maxIndex: vertices above it are dropped, edges if either id is, bearing-range factors if the pose id is."""
    graph, initial = NonlinearFactorGraph(), Values()
    lines = [ln.split() for ln in open(filename) if ln.strip()]
    for t in lines:
        if t[0] in ("VERTEX2", "VERTEX_SE2", "VERTEX"):
            if not max_index or int(t[1]) <= max_index:
                initial.insert_pose2(int(t[1]), float(t[2]), float(t[3]), float(t[4]))
        elif t[0] == "VERTEX_XY":
            if not max_index or int(t[1]) <= max_index:
                initial.insert_point2(L(int(t[1])), [float(t[2]), float(t[3])])
    for t in lines:
        if t[0] in ("EDGE2", "EDGE", "EDGE_SE2", "ODOMETRY"):
            id1, id2 = int(t[1]), int(t[2])
            if max_index and (id1 > max_index or id2 > max_index):
                continue
            x, y, yaw = float(t[3]), float(t[4]), float(t[5])
            graph.add_BetweenFactorPose2(id1, id2, [x, y, yaw], _noise2d(t[6:12], smart, noise_format))
            if not initial.exists(id1):
                initial.insert_pose2(id1, 0.0, 0.0, 0.0)
            if not initial.exists(id2):
                a = initial.at(id1)
                c, s = np.cos(a[2]), np.sin(a[2])
                initial.insert_pose2(id2, a[0] + c * x - s * y, a[1] + s * x + c * y, np.arctan2(np.sin(a[2] + yaw), np.cos(a[2] + yaw)))
        elif t[0] in ("BR", "LANDMARK"):
            id1, id2 = int(t[1]), int(t[2])
            if t[0] == "BR":
                bearing, rng, bearing_std, range_std = (float(x) for x in t[3:7])
            else:
                lmx, lmy, v1, _v2, v3 = (float(x) for x in t[3:8])
                bearing, rng = np.arctan2(lmy, lmx), np.sqrt(lmx * lmx + lmy * lmy)
                if abs(v1 - v3) < 1e-4:
                    bearing_std, range_std = np.sqrt(v1 / 10.0), np.sqrt(v1)
                else:
                    bearing_std, range_std = 1.0, 1.0
            if max_index and id1 > max_index:
                continue
            key2 = L(id2)
            graph.add_BearingRangeFactor2D(id1, key2, bearing, rng, noiseModel.Diagonal.Sigmas([bearing_std, range_std]))
            if not initial.exists(id1):
                initial.insert_pose2(id1, 0.0, 0.0, 0.0)
            if not initial.exists(key2):  # pose.transformFrom(bearing * Point2(range, 0))
                a = initial.at(id1)
                lx, ly = rng * np.cos(bearing), rng * np.sin(bearing)
                c, s = np.cos(a[2]), np.sin(a[2])
                initial.insert_point2(key2, [a[0] + c * lx - s * ly, a[1] + s * lx + c * ly])
    return graph, initial


def readG2o(filename, is3D=False):
    return load3D(filename) if is3D else load2D(filename, "g2o", True)


# ---------------------------------------------------------------- 3D
def _sym6(vals):
    m = np.zeros((6, 6))
    k = 0
    for i in range(6):
        for j in range(i, 6):
            m[i, j] = m[j, i] = float(vals[k])
            k += 1
    return m


def load3D(filename):
    """does NOT create missing vertices (dataset.cpp:929-938)."""
    graph, initial = NonlinearFactorGraph(), Values()
    for ln in open(filename):
        t = ln.split()
        if not t:
            continue
        if t[0] == "VERTEX3":
            x, y, z, roll, pitch, yaw = (float(a) for a in t[2:8])
            initial.insert_pose3(int(t[1]), rot3_ypr(yaw, pitch, roll), [x, y, z])
        elif t[0] == "VERTEX_SE3:QUAT":
            x, y, z, qx, qy, qz, qw = (float(a) for a in t[2:9])
            initial.insert_pose3(int(t[1]), rot3_quat(qw, qx, qy, qz), [x, y, z])
        elif t[0] == "VERTEX_TRACKXYZ":
            initial.insert_point3(symbol("l", int(t[1])), [float(a) for a in t[2:5]])
        elif t[0] == "EDGE3":
            id1, id2 = int(t[1]), int(t[2])
            x, y, z, roll, pitch, yaw = (float(a) for a in t[3:9])
            m = _sym6(t[9:30])
            graph.add_BetweenFactorPose3(id1, id2, rot3_ypr(yaw, pitch, roll), [x, y, z], noiseModel.Gaussian.Information(m))
        elif t[0] == "EDGE_SE3:QUAT":
            id1, id2 = int(t[1]), int(t[2])
            x, y, z, qx, qy, qz, qw = (float(a) for a in t[3:10])
            m = _sym6(t[10:31])
            mg = np.zeros((6, 6))
            mg[0:3, 0:3] = m[3:6, 3:6]
            mg[3:6, 3:6] = m[0:3, 0:3]
            mg[3:6, 0:3] = m[0:3, 3:6]
            mg[0:3, 3:6] = m[3:6, 0:3]
            graph.add_BetweenFactorPose3(id1, id2, rot3_quat(qw, qx, qy, qz), [x, y, z], noiseModel.Gaussian.Information(mg))
    return graph, initial


def chain_initial_pose3(graph: NonlinearFactorGraph, n_poses=None):
    """odometry-chained initial estimate for files without vertices (sphere2500.txt has EDGE3 lines only):
    pose 0 = identity, pose k+1 = pose k * measured(k, k+1) using the first edge (i, i+1) seen."""
    from .graph import F_BETWEEN_POSE3
    initial = Values()
    edges = {}
    for ftype, _, gi, keys, meas, _, _ in graph.buckets():
        if ftype != F_BETWEEN_POSE3:
            continue
        for k, m in zip(keys.tolist(), meas):
            if k[1] == k[0] + 1 and k[0] not in edges:
                edges[int(k[0])] = m
    R, t = np.eye(3), np.zeros(3)
    initial.insert_pose3(0, R, t)
    i = 0
    while i in edges and (n_poses is None or i + 1 < n_poses):
        m = edges[i]
        R, t = pose3_compose(R, t, m[:9].reshape(3, 3), m[9:12])
        i += 1
        initial.insert_pose3(i, R, t)
    return initial


# ---------------------------------------------------------------- writers (the formats either side of the hot path)
def _g(x):
    """C++ `stream << double` with the default precision 6 (what the reference's writeG2o uses)"""
    return "%g" % float(x)


def _information(model):
    """Info = R^T R of a Gaussian / Diagonal / Isotropic / Unit model (dataset.cpp:681-682)"""
    from .graph import N_DIAG, N_GAUSS, N_ISO, N_UNIT
    if model.kind == N_UNIT:
        return np.eye(model.dim)
    if model.kind in (N_ISO, N_DIAG):
        return np.diag(model.invsigmas() ** 2)
    assert model.kind == N_GAUSS
    R = model.data.reshape(model.dim, model.dim)
    return R.T @ R


def _rot3_to_quat(R):
    """(w, x, y, z), w >= 0 branch structure of Eigen's Quaternion(Matrix3)"""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        return 0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
    q = [0.0, 0.0, 0.0, 0.0]
    q[0] = (R[k, j] - R[j, k]) / s
    q[1 + i] = 0.25 * s
    q[1 + j] = (R[j, i] + R[i, j]) / s
    q[1 + k] = (R[k, i] + R[i, k]) / s
    return tuple(q)


def writeG2o(graph: NonlinearFactorGraph, estimate: Values, filename):
    """gtsam/slam/dataset.cpp:636-735: VERTEX_SE2 / VERTEX_SE3:QUAT / VERTEX_XY / VERTEX_TRACKXYZ then EDGE_SE2 / EDGE_SE3:QUAT with the upper
    triangle of the information matrix (3D: reordered to g2o's t,R convention); ids = Symbol(key).index()."""
    from .graph import F_BETWEEN_POSE2, F_BETWEEN_POSE3, POINT2, POINT3, POSE2, POSE3
    index = lambda key: int(key) & ((1 << 56) - 1)  # noqa: E731  Symbol(key).index()
    keys = sorted(estimate.keys())
    with open(filename, "w") as o:
        for k in keys:
            if estimate.type(k) == POSE2:
                v = estimate.at(k)
                o.write(f"VERTEX_SE2 {index(k)} {_g(v[0])} {_g(v[1])} {_g(v[2])}\n")
        for k in keys:
            if estimate.type(k) == POSE3:
                v = estimate.at(k)
                w, x, y, z = _rot3_to_quat(v[:9])
                o.write(f"VERTEX_SE3:QUAT {index(k)} {_g(v[9])} {_g(v[10])} {_g(v[11])} {_g(x)} {_g(y)} {_g(z)} {_g(w)}\n")
        for k in keys:
            if estimate.type(k) == POINT2:  # 2D landmarks, dataset.cpp:660-665
                v = estimate.at(k)
                o.write(f"VERTEX_XY {index(k)} {_g(v[0])} {_g(v[1])}\n")
        for k in keys:
            if estimate.type(k) == POINT3:
                v = estimate.at(k)
                o.write(f"VERTEX_TRACKXYZ {index(k)} {_g(v[0])} {_g(v[1])} {_g(v[2])}\n")
        rec = [None] * graph.size()
        for ftype, _, gi, fkeys, meas, _, models in graph.buckets():
            for i, g in enumerate(gi.tolist()):
                rec[g] = (ftype, fkeys[i], meas[i], models[i])
        for ftype, fk, m, model in rec:
            if ftype == F_BETWEEN_POSE2:
                info = _information(model)
                up = " ".join(_g(info[i, j]) for i in range(3) for j in range(i, 3))
                o.write(f"EDGE_SE2 {index(fk[0])} {index(fk[1])} {_g(m[0])} {_g(m[1])} {_g(m[2])} {up}\n")
            elif ftype == F_BETWEEN_POSE3:
                info = _information(model)
                ig = np.eye(6)
                ig[0:3, 0:3] = info[3:6, 3:6]
                ig[3:6, 3:6] = info[0:3, 0:3]
                ig[0:3, 3:6] = info[3:6, 0:3]
                ig[3:6, 0:3] = info[0:3, 3:6]
                w, x, y, z = _rot3_to_quat(m[:9])
                up = " ".join(_g(ig[i, j]) for i in range(6) for j in range(i, 6))
                o.write(f"EDGE_SE3:QUAT {index(fk[0])} {index(fk[1])} {_g(m[9])} {_g(m[10])} {_g(m[11])} {_g(x)} {_g(y)} {_g(z)} {_g(w)} {up}\n")


def _rot3_logmap(R):
    """SO3::Logmap (gtsam/geometry/SO3.cpp:299-375), the branch structure the oracle restates"""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    tr = np.trace(R)
    if tr + 1.0 < 1e-3:  # near pi: fall back on the largest-diagonal construction
        i = int(np.argmax(np.diag(R)))
        v = R[:, i].copy()
        v[i] += 1.0
        v *= np.pi / np.sqrt(2.0 * (1.0 + R[i, i]))
        return v
    tr_3 = tr - 3.0
    if tr_3 < -1e-6:
        theta = np.arccos(np.clip((tr - 1.0) / 2.0, -1.0, 1.0))
        magnitude = theta / (2.0 * np.sin(theta))
    else:
        magnitude = 0.5 - tr_3 / 12.0 + tr_3 * tr_3 / 60.0
    return magnitude * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])


def writeBAL(filename, db: SfmData):
    """gtsam/sfm/SfmData.cpp:249-323: precision 20, observations (camera, point, u, -v), cameras in the OpenGL convention
    (gtsam2openGL: R_gl = (wRc R90)^T ... written as Rodrigues vector, translation, f, k1, k2), then the points."""
    R90 = np.diag([1.0, -1.0, -1.0])
    n_obs = sum(len(t["measurements"]) for t in db.tracks)
    p = lambda x: "%.20g" % float(x)  # noqa: E731
    with open(filename, "w") as o:
        o.write(f"{len(db.cameras)} {len(db.tracks)} {n_obs}\n\n")
        for j, tr in enumerate(db.tracks):
            for i, (u, v) in tr["measurements"]:
                o.write(f"{i} {j} {p(u)} {p(-v)}\n")
        o.write("\n")
        for (wRc, wtc, f, k1, k2) in db.cameras:
            Rgl = (np.asarray(wRc) @ R90).T          # inverse of openGL2gtsam: R = (wRc R90)^T, t = -R wtc
            tgl = -Rgl @ np.asarray(wtc)
            w = _rot3_logmap(Rgl)
            o.write(f"{p(w[0])}\n{p(w[1])}\n{p(w[2])}\n{p(tgl[0])}\n{p(tgl[1])}\n{p(tgl[2])}\n{p(f)}\n{p(k1)}\n{p(k2)}\n\n")
        for tr in db.tracks:
            o.write(f"{p(tr['p'][0])}\n{p(tr['p'][1])}\n{p(tr['p'][2])}\n\n")
    return True
