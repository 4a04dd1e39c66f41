"""Synthetic BAL-shaped problems: cameras on a ring of radius 30 looking at the
origin, points uniform in [-8, 8]^3, every point seen by `obs_per_point` distinct random cameras, 0.5 px noise,
unit pixel noise model, priors on C(0) / P(0), perturbed initial estimate.  Pure numpy, seeded.

The draw order is this generator's own (numpy PCG64 seeded with `seed`): large instances are regenerated from
the seed on the GPU box rather than shipped."""
from __future__ import annotations

import numpy as np

from .graph import C, NonlinearFactorGraph, Ordering, P, Values, camera_pack, noiseModel


def _lookat(eye, target, up):
    """PinholeBase::LookatPose gtsam/geometry/CalibratedCamera.cpp:58-66"""
    zc = target - eye
    zc = zc / np.linalg.norm(zc)
    xc = np.cross(-up, zc)
    xc = xc / np.linalg.norm(xc)
    yc = np.cross(zc, xc)
    return np.stack([xc, yc, zc], axis=1), eye


def _expmap_pose(xi):
    """Pose3::Expmap (host copy used only to perturb the synthetic initial estimate)."""
    w, v = xi[:3], xi[3:]
    th2 = float(w @ w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th2 <= 1e-5:
        A, B, Cc = 1 - th2 / 6, 0.5 - th2 / 24, 1 / 6 - th2 / 120
    else:
        th = np.sqrt(th2)
        A = np.sin(th) / th
        B = 2 * np.sin(th / 2) ** 2 / th2
        Cc = (1 - A) / th2
    R = np.eye(3) + A * W + B * (W @ W)
    t = v + B * np.cross(w, v) + Cc * np.cross(w, np.cross(w, v))
    return R, t


def project_bundler(R, t, f, k1, k2, p):
    """vectorised PinholeCamera<Cal3Bundler>::project2 (value only): R (n,3,3), t (n,3), p (n,3)."""
    q = np.einsum("nji,nj->ni", R, p - t)
    u, v = q[:, 0] / q[:, 2], q[:, 1] / q[:, 2]
    r = u * u + v * v
    g = 1.0 + (k1 + k2 * r) * r
    return np.stack([f * g * u, f * g * v], axis=1), q[:, 2]


def make_bal(n_cam=20, n_pt=1000, obs_per_point=10, seed=42, pixel_sigma=0.5, with_priors=True, window=None):
    """returns (graph, initial, truth, ordering_schur).
    window = None: every point is seen by `obs_per_point` cameras drawn from ALL cameras (SURVEY's C4: every camera pair ends up
    co-visible, the camera block of the Hessian is dense).  window = w: the cameras of a point are drawn from w consecutive cameras of
    the ring around a random one -- the banded co-visibility of a real sequence (BAL's datasets: a camera shares points with its
    neighbours along the trajectory), whose nested-dissection ordering has small top separators over independent camera subtrees."""
    rng = np.random.default_rng(seed)
    ang = 2 * np.pi * np.arange(n_cam) / n_cam
    eyes = np.stack([30 * np.cos(ang), 30 * np.sin(ang), rng.uniform(-3, 3, n_cam)], axis=1)
    up = np.array([0.0, 0.0, 1.0])
    Rs, ts = np.zeros((n_cam, 3, 3)), np.zeros((n_cam, 3))
    for i in range(n_cam):
        Rs[i], ts[i] = _lookat(eyes[i], np.zeros(3), up)
    f = 500 + 20 * rng.uniform(-1, 1, n_cam)
    k1 = 1e-3 * rng.uniform(-1, 1, n_cam)
    k2 = 1e-4 * rng.uniform(-1, 1, n_cam)
    pts = rng.uniform(-8, 8, (n_pt, 3))
    k = min(obs_per_point, n_cam)
    if window is None:
        cam_idx = rng.integers(0, n_cam, (n_pt, k))
        redraw = lambda m: rng.integers(0, n_cam, (m, k))  # noqa: E731
    else:
        w = int(min(max(window, k), n_cam))
        centre = rng.integers(0, n_cam, n_pt)

        def draw(c):
            return (c[:, None] - w // 2 + rng.integers(0, w, (len(c), k))) % n_cam

        cam_idx = draw(centre)
    for _ in range(64):  # redraw rows that contain a repeated camera
        srt = np.sort(cam_idx, axis=1)
        bad = (srt[:, 1:] == srt[:, :-1]).any(axis=1)
        if not bad.any():
            break
        cam_idx[bad] = redraw(int(bad.sum())) if window is None else draw(centre[bad])
    cam_idx = np.sort(cam_idx, axis=1)
    ci = cam_idx.reshape(-1)
    pj = np.repeat(np.arange(n_pt), k)
    z, depth = project_bundler(Rs[ci], ts[ci], f[ci], k1[ci], k2[ci], pts[pj])
    assert (depth > 0).all()
    z = z + rng.normal(0, pixel_sigma, z.shape)

    graph = NonlinearFactorGraph()
    cam_keys = np.array([C(i) for i in range(n_cam)], dtype=np.uint64)
    pt_keys = (np.uint64(P(0)) + np.arange(n_pt, dtype=np.uint64))
    graph.add_GeneralSFMFactor(z, noiseModel.Isotropic.Sigma(2, 1.0), cam_keys[ci], pt_keys[pj])
    truth = Values()
    for i in range(n_cam):
        truth.insert_camera(int(cam_keys[i]), Rs[i], ts[i], f[i], k1[i], k2[i])
    for j in range(n_pt):
        truth.insert_point3(int(pt_keys[j]), pts[j])
    if with_priors:
        graph.add_PriorFactorCamera(C(0), truth.at(C(0)), noiseModel.Isotropic.Sigma(9, 0.1))
        graph.add_PriorFactorPoint3(P(0), truth.at(P(0)), noiseModel.Isotropic.Sigma(3, 0.1))
    initial = Values()
    for i in range(n_cam):
        xi = np.concatenate([rng.normal(0, 0.01, 3), rng.normal(0, 0.05, 3)])
        dR, dt = _expmap_pose(xi)
        initial.insert(int(cam_keys[i]), 3, camera_pack(Rs[i] @ dR, ts[i] + Rs[i] @ dt, f[i], k1[i], k2[i]))
    noise_p = rng.normal(0, 0.05, (n_pt, 3))
    for j in range(n_pt):
        initial.insert_point3(int(pt_keys[j]), pts[j] + noise_p[j])
    ordering = Ordering([int(x) for x in pt_keys] + [int(x) for x in cam_keys])
    return graph, initial, truth, ordering
