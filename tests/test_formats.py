"""Data formats either side of the hot path: writeG2o / writeBAL round trips through the loaders,
as gtsam/slam/tests/testDataset.cpp:writeG2o* and testSfmData.cpp:writeBAL* do (read -> write -> read, compare)."""
import os

import numpy as np

from gtsam_personal_amd.datasets import SfmData, load2D, load3D, readG2o, writeBAL, writeG2o

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _same_graph(g1, v1, g2, v2, tol):
    assert g1.size() == g2.size()
    b1 = {(b[0], b[1]): b for b in g1.buckets()}
    b2 = {(b[0], b[1]): b for b in g2.buckets()}
    assert b1.keys() == b2.keys()
    for k in b1:
        assert np.array_equal(b1[k][3], b2[k][3])                       # keys
        assert np.allclose(b1[k][4], b2[k][4], rtol=tol, atol=tol)      # measurements
        if b1[k][5] is not None:
            assert np.allclose(b1[k][5], b2[k][5], rtol=tol, atol=tol)  # whitening data
    assert sorted(v1.keys()) == sorted(v2.keys())
    for key in v1.keys():
        assert np.allclose(v1.at(key), v2.at(key), rtol=tol, atol=tol)


def test_write_g2o_2d_round_trip(tmp_path):
    g, v = readG2o(os.path.join(GOLD, "noisyToyGraph.txt"))
    out = str(tmp_path / "toy.g2o")
    writeG2o(g, v, out)
    g2, v2 = readG2o(out)
    _same_graph(g, v, g2, v2, 1e-5)  # the writer prints 6 significant digits like the reference's ostream default
    first = open(out).readline().split()
    assert first[0] == "VERTEX_SE2"


def test_write_g2o_3d_round_trip(tmp_path):
    g, v = load3D(os.path.join(GOLD, "pose3example.txt"))
    out = str(tmp_path / "p3.g2o")
    writeG2o(g, v, out)
    g2, v2 = load3D(out)
    _same_graph(g, v, g2, v2, 2e-5)
    tags = {ln.split()[0] for ln in open(out)}
    assert tags == {"VERTEX_SE3:QUAT", "EDGE_SE3:QUAT"}


def test_write_bal_round_trip(tmp_path):
    db = SfmData.FromBalFile(os.path.join(GOLD, "dubrovnik-3-7-pre.txt"))
    out = str(tmp_path / "d.txt")
    assert writeBAL(out, db)
    db2 = SfmData.FromBalFile(out)
    assert db2.numberCameras() == db.numberCameras() and db2.numberTracks() == db.numberTracks()
    for a, b in zip(db.cameras, db2.cameras):
        for x, y in zip(a, b):
            assert np.allclose(x, y, rtol=1e-5, atol=1e-5)  # the reader goes through float32 like the reference's
    for a, b in zip(db.tracks, db2.tracks):
        assert np.allclose(a["p"], b["p"], rtol=1e-6, atol=1e-6)
        assert [m[0] for m in a["measurements"]] == [m[0] for m in b["measurements"]]
        assert np.allclose([m[1] for m in a["measurements"]], [m[1] for m in b["measurements"]], rtol=1e-6, atol=1e-4)


def _quat_to_R(w, x, y, z):
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_load2d_toro_known_answers():
    """gtsam/slam/tests/testDataset.cpp:91-103 (w100.graph): 300 factors, 100 poses, first factor
    BetweenFactor<Pose2>(1, 0, Pose2(-0.99879, 0.0417574, -0.00818381), Unit(3))"""
    from gtsam_personal_amd.graph import N_UNIT
    g, v = load2D(os.path.join(GOLD, "w100.graph"))  # NoiseFormatAUTO -> GRAPH (covariance in TORO order), smart -> Unit
    assert g.size() == 300 and len(v.keys()) == 100
    (ftype, kind, gi, keys, meas, noise, models), = g.buckets()
    i0 = int(np.where(gi == 0)[0][0])
    assert keys[i0].tolist() == [1, 0]
    assert np.allclose(meas[i0], [-0.99879, 0.0417574, -0.00818381], atol=1e-9)
    assert models[i0].kind == N_UNIT


def test_read_g2o_2d_known_answers():
    """gtsam/slam/tests/testDataset.cpp:315-337 (pose2example.txt): Diagonal::Precisions(44.721360, 44.721360, 30.901699) and
    the eleven expected poses"""
    g, v = readG2o(os.path.join(GOLD, "pose2example.txt"))
    expected = [(0.0, 0.0, 0.0), (1.030390, 0.011350, -0.081596), (2.036137, -0.129733, -0.301887), (3.015097, -0.442395, -0.345514),
                (3.343949, 0.506678, 1.214715), (3.684491, 1.464049, 1.183785), (4.064626, 2.414783, 1.176333), (4.429778, 3.300180, 1.259169),
                (4.128877, 2.321481, -1.825391), (3.884653, 1.327509, -1.953016), (3.531067, 0.388263, -2.148934)]
    assert len(v.keys()) == 11
    for i, e in enumerate(expected):
        assert np.allclose(v.at(i), e, atol=1e-5)
    for b in g.buckets():
        for m in b[6]:
            assert np.allclose(m.invsigmas() ** 2, [44.721360, 44.721360, 30.901699], rtol=1e-6)


def test_read_g2o_3d_known_answers():
    """gtsam/slam/tests/testDataset.cpp:147-203 (pose3example.txt): six relative poses and five poses as (w, x, y, z) quaternion +
    translation, noise Isotropic::Precision(6, 10000), edges (0,1) (1,2) (2,3) (3,4) (1,4) (3,0)"""
    g, v = load3D(os.path.join(GOLD, "pose3example.txt"))
    rel = [((0.854230, 0.190253, 0.283162, -0.392318), (1.001367, 0.015390, 0.004948)),
           ((0.105373, 0.311512, 0.656877, -0.678505), (0.523923, 0.776654, 0.326659)),
           ((0.568551, 0.595795, -0.561677, 0.079353), (0.910927, 0.055169, -0.411761)),
           ((0.542221, -0.592077, 0.303380, -0.513226), (0.775288, 0.228798, -0.596923)),
           ((0.327419, -0.125250, -0.534379, 0.769122), (-0.577841, 0.628016, -0.543592)),
           ((0.083672, 0.104639, 0.627755, 0.766795), (-0.623267, 0.086928, 0.773222))]
    poses = [((1.0, 0.0, 0.0, 0.0), (0, 0, 0)), ((0.854230, 0.190253, 0.283162, -0.392318), (1.001367, 0.015390, 0.004948)),
             ((0.421446, -0.351729, -0.597838, 0.584174), (1.993500, 0.023275, 0.003793)),
             ((0.067024, 0.331798, -0.200659, 0.919323), (2.004291, 1.024305, 0.018047)),
             ((0.765488, -0.035697, -0.462490, 0.445933), (0.999908, 1.055073, 0.020212))]
    edges = [(0, 1), (1, 2), (2, 3), (3, 4), (1, 4), (3, 0)]
    (ftype, kind, gi, keys, meas, noise, models), = g.buckets()
    for i, ((q, t), e) in enumerate(zip(rel, edges)):
        k = int(np.where(gi == i)[0][0])
        assert tuple(keys[k].tolist()) == e
        assert np.allclose(meas[k][:9].reshape(3, 3), _quat_to_R(*q), atol=2e-5)
        assert np.allclose(meas[k][9:12], t, atol=1e-5)
        assert np.allclose(models[k].invsigmas() ** 2, 10000.0, rtol=1e-6)
    for j, (q, t) in enumerate(poses):
        assert np.allclose(v.at(j)[:9].reshape(3, 3), _quat_to_R(*q), atol=2e-5)
        assert np.allclose(v.at(j)[9:12], t, atol=1e-5)


def test_read_bal_dubrovnik_known_answers():
    """gtsam/sfm/tests/testSfmData.cpp:66-83: 3 cameras, 7 tracks, track 0 has 3 measurements, the first by camera 0, and camera 0
    projects track 0's point to within 12 px of that measurement"""
    from gtsam_personal_amd.synthetic import project_bundler
    db = SfmData.FromBalFile(os.path.join(GOLD, "dubrovnik-3-7-pre.txt"))
    assert db.numberCameras() == 3 and db.numberTracks() == 7
    tr = db.tracks[0]
    assert len(tr["measurements"]) == 3 and tr["measurements"][0][0] == 0
    R, t, f, k1, k2 = db.cameras[0]
    z, depth = project_bundler(np.asarray(R)[None], np.asarray(t)[None], np.array([f]), np.array([k1]), np.array([k2]), np.asarray(tr["p"])[None])
    assert depth[0] > 0
    assert np.abs(z[0] - np.array(tr["measurements"][0][1])).max() < 12


def test_load2d_victoria_park_counts_and_landmarks():
    """gtsam/slam/tests/testDataset.cpp:131-145 (load2DVictoriaPark): 10608 factors / 7120 values; restricted to maxIndex = 5:
    5 factors, 6 values, and the last factor's second key is L(5).  The file has ODOMETRY and LANDMARK lines only."""
    from gtsam_personal_amd.graph import F_BEARING_RANGE_2D, F_BETWEEN_POSE2, POINT2, POSE2, L
    path = os.path.join(GOLD, "victoria_park.txt")
    graph, initial = load2D(path)
    assert graph.size() == 10608 and initial.size() == 7120
    types = [initial.type(k) for k in initial.keys()]
    assert types.count(POSE2) == 6969 and types.count(POINT2) == 151
    counts = {ft: len(gi) for ft, _, gi, *_ in graph.buckets()}
    assert counts[F_BETWEEN_POSE2] == 6968 and counts[F_BEARING_RANGE_2D] == 3640
    graph2, initial2 = load2D(path, max_index=5)
    assert graph2.size() == 5 and initial2.size() == 6
    last = [(fk, ft) for ft, _, gi, fkeys, *_ in graph2.buckets() for g, fk in zip(gi.tolist(), fkeys) if g == 4]
    assert len(last) == 1 and last[0][1] == F_BEARING_RANGE_2D and int(last[0][0][1]) == L(5)
    # the first LANDMARK line: "LANDMARK 4 5 11.5387 -3.2007 0.4 0 0.4" -> bearing / range and sigmas (sqrt(0.04), sqrt(0.4))
    for ft, kind, gi, fkeys, meas, noise, models in graph2.buckets():
        if ft == F_BEARING_RANGE_2D:
            assert np.allclose(meas[0], [np.arctan2(-3.2007, 11.5387), np.hypot(11.5387, -3.2007)], atol=1e-12)
            assert np.allclose(models[0].data, [np.sqrt(0.04), np.sqrt(0.4)], atol=1e-12)
    # the landmark was not in the file as a vertex: it is initialised from the sighting, pose.transformFrom(bearing * (range, 0))
    p4, l5 = initial2.at(4), initial2.at(L(5))
    c, s = np.cos(p4[2]), np.sin(p4[2])
    assert np.allclose(l5, [p4[0] + c * 11.5387 - s * -3.2007, p4[1] + s * 11.5387 + c * -3.2007], atol=1e-9)


def test_writeg2o_roundtrips_planar_landmarks(tmp_path):
    from gtsam_personal_amd.graph import L
    graph, initial = load2D(os.path.join(GOLD, "victoria_park.txt"), max_index=40)
    out = tmp_path / "vp.g2o"
    writeG2o(graph, initial, str(out))
    text = out.read_text().splitlines()
    assert sum(ln.startswith("VERTEX_XY ") for ln in text) == sum(1 for k in initial.keys() if len(initial.at(k)) == 2)
    assert sum(ln.startswith("VERTEX_SE2 ") for ln in text) == sum(1 for k in initial.keys() if len(initial.at(k)) == 3)
    _, again = load2D(str(out), "g2o")
    for k in initial.keys():
        kk = k if len(initial.at(k)) == 3 else L(int(k) & ((1 << 56) - 1))
        assert np.allclose(again.at(kk), initial.at(k), rtol=1e-5, atol=1e-6)
