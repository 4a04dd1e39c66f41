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


def stale_landmark_steps(n_poses=14):
    """poses on a line with exact odometry (their deltas stay ~0); a landmark seen twice from pose 2 with conflicting measurements: its
    delta stays large until the first relinearization check, by which time its clique lies below cliques of unmoved poses -- the case in
    which CheckRelinearizationPartial (ISAM2-impl.h:302-331) marks less than CheckRelinearizationFull"""
    odo = noiseModel.Diagonal.Sigmas([0.05, 0.05, 0.02])
    br = noiseModel.Diagonal.Sigmas([0.05, 0.1])
    steps = []
    for i in range(n_poses):
        g, v = NonlinearFactorGraph(), Values()
        v.insert_pose2(i, float(i), 0.0, 0.0)
        if i == 0:
            g.add_PriorFactorPose2(0, [0.0, 0.0, 0.0], odo)
        else:
            g.add_BetweenFactorPose2(i - 1, i, [1.0, 0.0, 0.0], odo)
        if i == 2:
            g.add_BearingRangeFactor2D(2, 100, np.pi / 2, 5.0, br)
            v.insert_point2(100, [2.0, 5.0])
        if i == 3:
            g.add_BearingRangeFactor2D(2, 100, np.pi / 2 + 0.2, 6.0, br)
        steps.append((g, v))
    return steps


from gtsam_personal_amd.incremental_workloads import (create_points, create_poses, incremental_pose2_steps, project_cal3_s2, visual_steps,  # noqa: E402,F401
                                                    write_isam2_sequence)
