"""Data formats either side of the hot path: writeG2o / writeBAL round trips through the loaders,
as gtsam/slam/tests/testDataset.cpp:writeG2o* and testSfmData.cpp:writeBAL* do (read -> write -> read, compare)."""
import os

import numpy as np

from gtsam_personal_amd.datasets import SfmData, load2D, load3D, writeBAL, writeG2o

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
    g, v = load2D(os.path.join(GOLD, "noisyToyGraph.txt"))
    out = str(tmp_path / "toy.g2o")
    writeG2o(g, v, out)
    g2, v2 = load2D(out)
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
