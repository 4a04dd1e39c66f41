"""CPU side of the headline-workload fixtures (tests/tools/make_c4_fixture.py): the small fixture equals what the oracle computes
live (so the committed numbers are not stale with respect to the oracle or the generator), and the product's symbolic analysis
reproduces the fixture's structure at FULL size under the reference's METIS ordering (no GPU needed: structure-only handle)."""
import os

import numpy as np
import pytest

import oracle_harness as oh
from gtsam_personal_amd import LevenbergMarquardtOptimizer, LevenbergMarquardtParams
from gtsam_personal_amd.synthetic import make_bal

from test_gpu_c4 import CASES, GOLD, fixture_ordering


def test_small_fixture_matches_live_oracle():
    fx = np.load(os.path.join(GOLD, "bal100_seed42_metis.npz"))
    n_cam, n_pt, obs, seed = CASES["bal100_seed42"]
    graph, initial, _, schur = make_bal(n_cam, n_pt, obs, seed=seed)
    ordering = fixture_ordering(fx, schur)
    if oh.have_ref():  # the permutation itself, recomputed with the reference's METIS (oracle/_ref)
        assert oh.metis(graph) == ordering
    params = LevenbergMarquardtParams()
    orc = oh.OracleProblem(graph, initial, ordering)
    orc.lm_init(params)
    assert orc.lm_state()["error"] == float(fx["error_initial"])
    orc.lm_iterate(params)
    st = orc.lm_state()
    assert st["error"] == float(fx["error_after"]) and st["inner"] == int(fx["inner"])
    d = orc.get_delta()
    assert np.linalg.norm(d) == float(fx["delta_norm"])


@pytest.mark.parametrize("ordering_name", ["schur", "metis"])
def test_full_size_structure_matches_fixture(ordering_name):
    fx = np.load(os.path.join(GOLD, f"c4_seed42_{ordering_name}.npz"))
    n_cam, n_pt, obs, seed = CASES["c4_seed42"]
    graph, initial, _, schur = make_bal(n_cam, n_pt, obs, seed=seed)
    opt = LevenbergMarquardtOptimizer(graph, initial, fixture_ordering(fx, schur), device=-1)
    assert opt.num_fronts() == int(fx["num_cliques"])
    root = opt.num_fronts() - 1
    info = opt.front_info(root)
    assert (info["nf"], info["n"]) == tuple(int(x) for x in fx["root_shape"])
    rkeys, _ = opt.front(root, numeric=False)
    assert rkeys == [int(k) for k in fx["root_keys"]]
    for j, ci in enumerate(fx["leaf_ids"].tolist()):
        fk, _ = opt.front(ci, numeric=False)
        assert fk == [int(k) for k in fx[f"leaf{j}_keys"]], ci
