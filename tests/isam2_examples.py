"""The reference's own incremental workloads as sequences of ISAM2::update inputs, restated with the Python mirror's graph types:
  slamlike_steps()   createSlamlikeISAM2, tests/testGaussianISAM2.cpp:44-168 (Pose2 odometry + two bearing-range landmarks)
  visual_steps()     examples/VisualISAM2Example.cpp:58-141 = tests/testVisualISAM2.cpp:33-118 with examples/SFMdata.h:42-76
Each step is (newFactors, newValues); an empty pair stands for the bare `isam.update()` the visual example issues."""
import numpy as np

from gtsam_personal_amd import NonlinearFactorGraph, Values, noiseModel
from gtsam_personal_amd.datasets import pose3_compose, rot3_expmap, rot3_ypr
from gtsam_personal_amd.graph import symbol


def slamlike_steps(max_poses=10):
    """[(graph, values)] of createSlamlikeISAM2(maxPoses); the `goto done` logic of the reference is kept"""
    odo = noiseModel.Diagonal.Sigmas([0.1, 0.1, np.pi / 100.0])
    br = noiseModel.Diagonal.Sigmas([np.pi / 100.0, 0.1])
    steps = []

    def step(build):
        g, v = NonlinearFactorGraph(), Values()
        build(g, v)
        steps.append((g, v))

    i = 0
    step(lambda g, v: (g.add_PriorFactorPose2(0, [0.0, 0.0, 0.0], odo), v.insert_pose2(0, 0.01, 0.01, 0.01)))
    if i > max_poses:
        return steps
    while i < 5:
        step(lambda g, v, i=i: (g.add_BetweenFactorPose2(i, i + 1, [1.0, 0.0, 0.0], odo), v.insert_pose2(i + 1, float(i + 1) + 0.1, -0.1, 0.01)))
        if i > max_poses:
            return steps
        i += 1
    if i > max_poses:
        return steps

    def lm1(g, v, i=i):
        g.add_BetweenFactorPose2(i, i + 1, [1.0, 0.0, 0.0], odo)
        g.add_BearingRangeFactor2D(i, 100, np.pi / 4.0, 5.0, br)
        g.add_BearingRangeFactor2D(i, 101, -np.pi / 4.0, 5.0, br)
        v.insert_pose2(i + 1, 1.01, 0.01, 0.01)
        v.insert_point2(100, [5.0 / np.sqrt(2.0), 5.0 / np.sqrt(2.0)])
        v.insert_point2(101, [5.0 / np.sqrt(2.0), -5.0 / np.sqrt(2.0)])
    step(lm1)
    i += 1
    if i > max_poses:
        return steps
    while i < 10:
        step(lambda g, v, i=i: (g.add_BetweenFactorPose2(i, i + 1, [1.0, 0.0, 0.0], odo), v.insert_pose2(i + 1, float(i + 1) + 0.1, -0.1, 0.01)))
        if i > max_poses:
            return steps
        i += 1
    if i > max_poses:
        return steps

    def lm2(g, v, i=i):
        g.add_BetweenFactorPose2(i, i + 1, [1.0, 0.0, 0.0], odo)
        g.add_BearingRangeFactor2D(i, 100, np.pi / 4.0 + np.pi / 16.0, 4.5, br)
        g.add_BearingRangeFactor2D(i, 101, -np.pi / 4.0 + np.pi / 16.0, 4.5, br)
        v.insert_pose2(i + 1, 6.9, 0.1, 0.01)
    step(lm2)
    return steps


def constrained_ordering_steps():
    """TEST(ISAM2, constrained_ordering), tests/testGaussianISAM2.cpp:479-571: the slamlike sequence, from the fourth odometry step on
    every update is given constrainedKeys {3: 1, 4: 2}.  [(graph, values, constrainedKeys | None)]"""
    constrained = {3: 1, 4: 2}
    out = []
    for n, (g, v) in enumerate(slamlike_steps()):
        # steps: 0 = prior, 1..5 = odometry i = 0..4 (constrained for i >= 3), 6.. = everything after, all constrained
        out.append((g, v, constrained if n >= 4 else None))
    return out


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
