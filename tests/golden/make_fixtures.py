"""Regenerates the data fixtures in this directory from the reference's example data (run in the build
container only; the fixtures themselves are committed).  Fixtures are DATA: input files the reference's own
tests/examples read (examples/Data/*), some cut down so that the CPU oracle finishes in seconds."""
import os
import shutil

REF = "/root/reference/examples/Data"
HERE = os.path.dirname(os.path.abspath(__file__))

# verbatim inputs (sphere2500 / city10000 in full: configs C3 / C1 at the reference's own size)
# (victoria_park.txt: the planar landmark-SLAM input of timing/timeIncremental.cpp -- bearing-range factors, 6 969 poses)
for name in ("dubrovnik-3-7-pre.txt", "pose3example.txt", "noisyToyGraph.txt", "sphere2500.txt", "city10000.g2o", "w100.graph", "pose2example.txt",
             "victoria_park.txt"):
    shutil.copy(os.path.join(REF, name), os.path.join(HERE, name))

# first 300 poses of sphere2500 (EDGE3 lines whose two ids are < 300)
with open(os.path.join(REF, "sphere2500.txt")) as f, open(os.path.join(HERE, "sphere2500_head.txt"), "w") as o:
    for ln in f:
        t = ln.split()
        if t and t[0] == "EDGE3" and int(t[1]) < 300 and int(t[2]) < 300:
            o.write(ln)

# first 400 poses of city10000.g2o
with open(os.path.join(REF, "city10000.g2o")) as f, open(os.path.join(HERE, "city10000_head.g2o"), "w") as o:
    for ln in f:
        t = ln.split()
        if not t:
            continue
        if t[0] == "VERTEX_SE2" and int(t[1]) < 400:
            o.write(ln)
        if t[0] == "EDGE_SE2" and int(t[1]) < 400 and int(t[2]) < 400:
            o.write(ln)

# Derived fixtures (numbers the oracle / the reference's vendored CCOLAMD + METIS produced; each has its own generator):
#   c4_seed42_{schur,metis}.npz, bal100_seed42_{schur,metis}.npz, *_timing.json   tests/tools/make_c4_fixture.py
#   slam_orderings.npz                                                             tests/tools/make_ordering_fixtures.py
