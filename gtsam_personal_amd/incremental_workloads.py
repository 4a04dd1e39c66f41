"""The reference's incremental workloads as sequences of ISAM2::update inputs (BASELINE.json configs[4] and the reference's own incremental
benchmark), built with the package's graph types -- the workload definitions bench.py --workload isam2 and the tests share:
  visual_steps()             examples/VisualISAM2Example.cpp:58-141 (= tests/testVisualISAM2.cpp:33-118) with examples/SFMdata.h:42-76
  incremental_pose2_steps()  timing/timeIncremental.cpp:84-170 on a Pose2 g2o file: one pose per update
  write_isam2_sequence()     the neutral text form of such a sequence that the C++ driver over the C ABI reads (tests/cpp/isam2_harness.cpp)
Each step is (newFactors, newValues); an empty pair stands for the bare `isam.update()` the visual example issues."""
import numpy as np

from .datasets import pose3_compose, rot3_expmap, rot3_ypr
from .graph import NonlinearFactorGraph, Values, noiseModel, symbol


def create_points():
    return [np.array(p, dtype=np.float64) for p in ((10, 10, 10), (-10, 10, 10), (-10, -10, 10), (10, -10, 10), (10, 10, -10), (-10, 10, -10),
                                                     (-10, -10, -10), (10, -10, -10))]


def create_poses(steps=8):
    """examples/SFMdata.h:61-76: circular trajectory of radius 30, always facing the centre"""
    R, t = rot3_ypr(np.pi / 2, 0.0, -np.pi / 2), np.array([30.0, 0.0, 0.0])
    dR, dt = rot3_ypr(0.0, -np.pi / 4, 0.0), np.array([np.sin(np.pi / 4) * 30, 0.0, 30 * (1 - np.sin(np.pi / 4))])
    poses = [(R, t)]
    for _ in range(1, steps):
        R, t = pose3_compose(R, t, dR, dt)
        poses.append((R, t))
    return poses


def project_cal3_s2(R, t, p, K):
    q = R.T @ (p - t)
    u, v = q[0] / q[2], q[1] / q[2]
    fx, fy, s, u0, v0 = K
    return np.array([fx * u + s * v + u0, fy * v + v0])


def visual_steps(extra_update=True):
    """[(graph, values)]: frame 0 is held back and goes in with frame 1 (VisualISAM2Example.cpp:103-127); after every update the
    example calls a bare isam.update() once more"""
    K = (50.0, 50.0, 0.0, 50.0, 50.0)
    noise = noiseModel.Isotropic.Sigma(2, 1.0)
    points, poses = create_points(), create_poses()
    dR, dt = rot3_expmap([-0.1, 0.2, 0.25]), np.array([0.05, -0.10, 0.20])
    steps = []
    g, v = NonlinearFactorGraph(), Values()
    for i, (R, t) in enumerate(poses):
        for j, p in enumerate(points):
            g.add_GenericProjectionFactor(project_cal3_s2(R, t, p, K), noise, symbol("x", i), symbol("l", j), K)
        Ri, ti = pose3_compose(R, t, dR, dt)
        v.insert_pose3(symbol("x", i), Ri, ti)
        if i == 0:
            g.add_PriorFactorPose3(symbol("x", 0), R, t, noiseModel.Diagonal.Sigmas([0.1, 0.1, 0.1, 0.3, 0.3, 0.3]))
            g.add_PriorFactorPoint3(symbol("l", 0), points[0], noiseModel.Isotropic.Sigma(3, 0.1))
            for j, p in enumerate(points):
                v.insert_point3(symbol("l", j), p + np.array([-0.25, 0.20, 0.15]))
        else:
            steps.append((g, v))
            if extra_update:
                steps.append((NonlinearFactorGraph(), Values()))
            g, v = NonlinearFactorGraph(), Values()
    return steps


def incremental_pose2_steps(g2o_path, n_poses, init_from):
    """timing/timeIncremental.cpp:84-170 on a Pose2 g2o file: one pose per update with the edges that reach back from it; the new pose is
    the previous one's estimate (init_from(step - 1) -> 3-vector, asked before the update that adds pose `step`) composed with the
    odometry.  A generator of (graph, values): the caller applies each update before asking for the next."""
    from .datasets import readG2o
    graph, _ = readG2o(g2o_path)
    edges = []
    for ftype, kind, gi, keys, meas, noise, models in graph.buckets():
        for i, g in enumerate(gi.tolist()):
            edges.append((g, int(keys[i][0]), int(keys[i][1]), meas[i], models[i]))
    edges.sort()

    def compose(a, d):
        c, s = np.cos(a[2]), np.sin(a[2])
        return np.array([a[0] + c * d[0] - s * d[1], a[1] + s * d[0] + c * d[1], a[2] + d[2]])

    nxt, step = 0, 1
    while nxt < len(edges) and step <= n_poses:
        g, v = NonlinearFactorGraph(), Values()
        if step == 1:
            v.insert_pose2(0, 0.0, 0.0, 0.0)
            g.add_PriorFactorPose2(0, [0.0, 0.0, 0.0], noiseModel.Unit.Create(3))
        while nxt < len(edges):
            _, k1, k2, m, model = edges[nxt]
            if k1 > step or k2 > step:
                break
            g.add_BetweenFactorPose2(k1, k2, m, model)
            if k2 == step and k1 == step - 1:
                v.insert(step, 0, compose(np.zeros(3) if step == 1 else init_from(step - 1), m))
            nxt += 1
        yield g, v
        step += 1


def fixed_lag_pose2_steps(g2o_path, n_poses, lag, init_from):
    """A fixed-lag smoother's use of ISAM2 (gtsam_unstable/nonlinear/IncrementalFixedLagSmoother.cpp: order the keys that are about to
    leave first, update, ISAM2::marginalizeLeaves) on a Pose2 g2o file, one pose per update: after the update that adds pose s, pose
    s - lag is marginalized out.  Edges that reach back beyond the window are dropped (their older end has left the system).
    Yields (graph, values, leaving_keys); init_from as in incremental_pose2_steps."""
    for step, (g, v) in enumerate(incremental_pose2_steps(g2o_path, n_poses, init_from), start=1):
        gw = NonlinearFactorGraph()
        for ftype, kind, gi, keys, meas, noise, models in g.buckets():
            order = np.argsort(gi)
            for i in order.tolist():
                if len(keys[i]) > 1 and ftype == 1 and min(int(keys[i][0]), int(keys[i][1])) < step - lag:
                    continue
                if ftype == 1:
                    gw.add_BetweenFactorPose2(int(keys[i][0]), int(keys[i][1]), meas[i], models[i])
                else:
                    gw.add_PriorFactorPose2(int(keys[i][0]), meas[i], models[i])
        yield gw, v, ([step - lag] if step - lag >= 0 else [])


def fixed_lag_update_params(cliques, existing_keys, new_keys, leaving):
    """what IncrementalFixedLagSmoother hands to ISAM2::update for keys that are about to be marginalized (the reference's own test helper
    does the same: tests/testGaussianISAM2.cpp:662-726): constrainedKeys = leaving keys in group 0, everything else in group 1;
    extraReelimKeys = the leaving keys and the frontals of every clique below them that has them in its separator.
    cliques: [(keys, n_frontal_keys, RSd, parent)] depth-first (the tree taps).  Returns (constrainedKeys, extraReelimKeys)."""
    if not leaving:
        return None, []
    constrained = {int(k): 1 for k in existing_keys}
    for k in new_keys:
        constrained[int(k)] = 1
    for k in leaving:
        constrained[int(k)] = 0
    children = {i: [] for i in range(len(cliques))}
    for i, (_, _, _, par) in enumerate(cliques):
        if par >= 0:
            children[par].append(i)
    marked = []
    for key in sorted(leaving):
        marked.append(int(key))
        home = next(i for i, (keys, nf, _, _) in enumerate(cliques) if key in keys[:nf])
        stack = list(children[home])
        while stack:
            i = stack.pop()
            keys, nf, _, _ = cliques[i]
            if key in keys[nf:]:
                marked.extend(int(k) for k in keys[:nf])
                stack.extend(children[i])
    return constrained, marked


def write_isam2_sequence(path, params, steps, relative_pose2=False):
    """the input of tests/cpp/isam2_harness: what the reference-side wrapper extracts from each update's NonlinearFactorGraph / Values
    (the packings of include/lmgpu.h), as text.  steps: [(graph, values[, removeFactorIndices[, {"constrained": {key: group},
    "extra_reelim": [keys], "marginalize": [keys]}]])].  relative_pose2: a new Pose2 k that
    comes with a BetweenFactor<Pose2>(k - 1, k) is written as "previous estimate composed with that odometry" (W line): the harness asks
    the device for calculateEstimate(k - 1) at update time, like timing/timeIncremental.cpp"""
    from .graph import FACTOR_ARITY, F_BETWEEN_POSE2, F_PRIOR_CAM, F_SFM, N_UNIT, POSE2, VAR_STORE_DEV
    u0v0 = {}
    with open(path, "w") as f:
        p = params
        f.write(f"ISAM2 {p.relinearizeThreshold!r} {int(p.relinearizeSkip)} {int(bool(p.enableRelinearization))} {p.optimizationParams.wildfireThreshold!r}\n")
        if getattr(p, "findUnusedFactorSlots", False):
            f.write("OPT find_unused_slots 1\n")
        for st in steps:
            g, v = st[0], st[1]
            rm = list(st[2]) if len(st) > 2 and st[2] is not None else []
            f.write(f"UPDATE {v.size()} {g.size()} {len(rm)}\n")
            odo = {}
            if relative_pose2:
                for ftype, _, _, keys, meas, _, _ in g.buckets():
                    if ftype == F_BETWEEN_POSE2:
                        for i in range(len(keys)):
                            odo[(int(keys[i][0]), int(keys[i][1]))] = meas[i]
            for k in v.keys():
                t = v.type(k)
                if t == POSE2 and (int(k) - 1, int(k)) in odo:
                    m = odo[(int(k) - 1, int(k))]
                    f.write(f"W {int(k)} {int(k) - 1} {float(m[0])!r} {float(m[1])!r} {float(m[2])!r}\n")
                    continue
                if t == 3:
                    u0v0[k] = v.at(k)[15:17].copy()
                f.write(f"V {int(k)} {t} {VAR_STORE_DEV[t]} " + " ".join(repr(float(x)) for x in v.at(k)[:VAR_STORE_DEV[t]]) + "\n")
            rows = [None] * g.size()
            for ftype, kind, gi, keys, meas, noise, models in g.buckets():
                for i, gidx in enumerate(gi.tolist()):
                    m = np.array(meas[i], dtype=np.float64)
                    if ftype == F_SFM:
                        m = m - u0v0[int(keys[i][0])]
                    if ftype == F_PRIOR_CAM:
                        m = m[:15]
                    kk = [int(x) for x in keys[i][:FACTOR_ARITY[ftype]]] + [0] * (3 - FACTOR_ARITY[ftype])
                    nz = [] if kind == N_UNIT else np.asarray(noise[i], dtype=np.float64).reshape(-1).tolist()
                    rows[gidx] = (f"F {ftype} {kk[0]} {kk[1]} {kk[2]} {len(m)} " + " ".join(repr(float(x)) for x in m) + f" {kind} {len(nz)} " +
                                  " ".join(repr(float(x)) for x in nz) + "\n")
            f.writelines(rows)
            f.write("R " + " ".join(str(int(i)) for i in rm) + "\n")
            extra = st[3] if len(st) > 3 and st[3] else {}
            if extra.get("constrained") is not None:  # ISAM2UpdateParams::constrainedKeys: key group pairs
                ck = sorted(extra["constrained"].items())
                f.write(f"C {len(ck)} " + " ".join(f"{int(k)} {int(g_)}" for k, g_ in ck) + "\n")
            if extra.get("extra_reelim"):  # ISAM2UpdateParams::extraReelimKeys
                f.write(f"X {len(extra['extra_reelim'])} " + " ".join(str(int(k)) for k in extra["extra_reelim"]) + "\n")
            if extra.get("marginalize"):  # ISAM2::marginalizeLeaves after the update
                f.write(f"M {len(extra['marginalize'])} " + " ".join(str(int(k)) for k in extra["marginalize"]) + "\n")
        f.write("END\n")
