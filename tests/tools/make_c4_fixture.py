#!/usr/bin/env python3
"""BUILD-CONTAINER ONLY: runs the CPU oracle (oracle/liblm_oracle.so, 1 thread) ONCE on the headline workload
BASELINE.json configs[3] — synthetic BAL 1 000 cameras / 100 000 points / 1 000 002 factors, seed 42 — for ONE
LevenbergMarquardtOptimizer::iterate() under the Schur ordering and under the reference's METIS ordering
(Ordering::Metis, gtsam/inference/Ordering.cpp:211-256, through oracle/_ref/libmetis_ref.so), and writes what the GPU
parity test and bench.py need as small fixtures:

  tests/golden/c4_seed42_<ordering>.npz   initial error, error after the iteration, lambda, inner iterations, norm of the
                                          update vector, all camera entries and sampled point entries of delta (by key),
                                          the root clique's key order, sampled entries of the root's [R S d], a few whole
                                          point cliques; for METIS additionally the permutation itself
  tests/golden/c4_seed42_timing.json      wall time of the oracle's iteration at FULL size (cpu_baseline at the same workload)

Usage: python tests/tools/make_c4_fixture.py [--cams 1000 --points 100000 --obs 10 --seed 42] [--orderings schur,metis] [--tag c4_seed42]
The same script with smaller sizes (--tag bal100_seed42 --cams 100 --points 10000) makes the 1/10-scale fixture."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_harness as oh  # noqa: E402
from gtsam_personal_amd import LevenbergMarquardtParams  # noqa: E402
from gtsam_personal_amd.synthetic import make_bal  # noqa: E402


def metis_ordering_fast(graph):
    """Ordering::Metis for a graph of binary (camera, point) + unary factors without Python-level sets: MetisIndex numbers
    the keys in order of first appearance and lists each key's neighbours ascending (MetisIndex-inl.h:27-82)."""
    fk = graph.factor_keys_in_graph_order()
    int_of, keys = {}, []
    a, b = [], []
    for ks in fk:
        for k in ks:
            if k not in int_of:
                int_of[k] = len(keys)
                keys.append(k)
        if len(ks) == 2:
            a.append(int_of[ks[0]])
            b.append(int_of[ks[1]])
        elif len(ks) > 2:
            raise ValueError("binary factors only")
    a, b = np.array(a, dtype=np.int64), np.array(b, dtype=np.int64)
    src = np.concatenate([a, b])
    dst = np.concatenate([b, a])
    pair = np.unique(src * len(keys) + dst)
    src, dst = pair // len(keys), pair % len(keys)
    present = np.unique(src)
    assert len(present) == len(keys), "a key without neighbour: MetisIndex gives it no row (not handled by the fast path)"
    xadj = np.concatenate([[0], np.cumsum(np.bincount(src, minlength=len(keys)))])
    perm, _ = oh.metis_from_adjacency(xadj, dst)
    return [keys[i] for i in perm]


def run(graph, initial, ordering, name, out_path, rng):
    params = LevenbergMarquardtParams()
    t0 = time.perf_counter()
    orc = oh.OracleProblem(graph, initial, ordering)
    t_build = time.perf_counter() - t0
    orc.lm_init(params)
    e0 = orc.lm_state()["error"]
    t0 = time.perf_counter()
    orc.lm_iterate(params)
    t_iter = time.perf_counter() - t0
    st = orc.lm_state()
    tm = orc.timings()
    trace = orc.lm_trace()
    delta = orc.get_delta()  # ordering order
    keys = np.array(list(ordering), dtype=np.uint64)
    types = np.array([initial.type(int(k)) for k in keys])
    dims = np.where(types == 3, 9, 3)
    off = np.concatenate([[0], np.cumsum(dims)])
    is_cam = types == 3
    cam_keys = keys[is_cam]
    cam_delta = np.stack([delta[off[i]:off[i] + 9] for i in np.nonzero(is_cam)[0]])
    pt_idx = rng.choice(np.nonzero(~is_cam)[0], size=min(2000, int((~is_cam).sum())), replace=False)
    pt_keys = keys[pt_idx]
    pt_delta = np.stack([delta[off[i]:off[i] + 3] for i in pt_idx])
    L = orc.L
    ncl = L.orc_num_cliques(orc.h)
    info = np.zeros(5, dtype=np.int32)
    L.orc_clique_info(orc.h, ncl - 1, oh.ip(info))
    root_keys = np.zeros(info[0], dtype=np.uint64)
    root = np.empty(int(info[2]) * int(info[3]))
    L.orc_clique_get(orc.h, ncl - 1, oh.up(root_keys), oh.dp(root))
    nf, n = int(info[2]), int(info[3])
    root = root.reshape(n, nf).T  # (nf, n)
    rr = np.concatenate([rng.integers(0, nf, 3000), [0, 1, 63, 64, 255, 256, nf - 1, nf - 2]])
    cc = np.array([rng.integers(r, n) for r in rr])
    cc[-8:] = [0, 1, 63, 64, 255, 256, nf - 1, n - 1]
    rr = np.concatenate([rr, np.arange(0, nf, max(1, nf // 500))])  # + entries of d (the right-hand-side column)
    cc = np.concatenate([cc, np.full(len(rr) - len(cc), n - 1)])
    root_vals = root[rr, cc]
    root_diag = np.diag(root[:, :nf]).copy()
    # a few whole point cliques
    leaf_ids = rng.choice(ncl - 1, size=min(24, ncl - 1), replace=False)
    leaf = {}
    for j, ci in enumerate(leaf_ids.tolist()):
        L.orc_clique_info(orc.h, ci, oh.ip(info))
        k = np.zeros(info[0], dtype=np.uint64)
        m = np.empty(int(info[2]) * int(info[3]))
        L.orc_clique_get(orc.h, ci, oh.up(k), oh.dp(m))
        leaf[f"leaf{j}_keys"] = k
        leaf[f"leaf{j}_rsd"] = m.reshape(int(info[3]), int(info[2])).T.copy()
        leaf[f"leaf{j}_meta"] = np.array([ci, info[1], info[4]])
    out = dict(error_initial=e0, error_after=st["error"], lambda_after=st["lambda_"], inner=st["inner"], iterations=st["iterations"],
               trace=trace, delta_norm=np.linalg.norm(delta), cam_keys=cam_keys, cam_delta=cam_delta, pt_keys=pt_keys, pt_delta=pt_delta,
               num_cliques=ncl, root_keys=root_keys, root_shape=np.array([nf, n]), root_rows=rr, root_cols=cc, root_vals=root_vals,
               root_diag=root_diag, leaf_ids=leaf_ids, **leaf)
    if name != "schur":
        srt = np.sort(keys)
        out["ordering_perm"] = np.searchsorted(srt, keys).astype(np.int32)  # ordering = sorted(keys)[perm]
    np.savez_compressed(out_path, **out)
    timing = dict(ordering=name, build_s=t_build, iterate_s=t_iter, linearize_s=tm["linearize_s"], eliminate_s=tm["eliminate_s"],
                  backsub_s=tm["backsub_s"], inner=st["inner"], error_initial=e0, error_after=st["error"])
    print(json.dumps(timing), flush=True)
    return timing


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cams", type=int, default=1000)
    ap.add_argument("--points", type=int, default=100000)
    ap.add_argument("--obs", type=int, default=10)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--orderings", default="schur,metis")
    ap.add_argument("--tag", default="c4_seed42")
    ap.add_argument("--window", type=int, default=None, help="banded co-visibility (synthetic.make_bal window=): the multi-GPU camera-subtree workload")
    args = ap.parse_args()
    graph, initial, _, schur = make_bal(args.cams, args.points, args.obs, seed=args.seed, window=args.window)
    gold = os.path.join(ROOT, "tests", "golden")
    timings = []
    for name in args.orderings.split(","):
        rng = np.random.default_rng(1234)
        if name == "schur":
            ordering = list(schur)
        elif name == "metis":
            t0 = time.perf_counter()
            ordering = metis_ordering_fast(graph)
            print(f"METIS ordering: {time.perf_counter() - t0:.1f} s", flush=True)
        else:
            raise SystemExit(name)
        timings.append(run(graph, initial, ordering, name, os.path.join(gold, f"{args.tag}_{name}.npz"), rng))
    import platform
    with open(os.path.join(gold, f"{args.tag}_timing.json"), "w") as f:
        json.dump(dict(workload=f"synthetic BAL {args.cams} cameras / {args.points} points / {graph.size()} factors, seed {args.seed}" +
                                (f", co-visibility window {args.window}" if args.window else ""),
                       what="oracle/liblm_oracle.so (CPU restatement of the reference's algorithm), ONE LevenbergMarquardtOptimizer::iterate(), "
                            "1 thread, build container (8 vCPU Xeon 2.1 GHz)", host=platform.processor() or platform.machine(),
                       runs=timings), f, indent=1)


if __name__ == "__main__":
    main()
