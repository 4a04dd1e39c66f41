"""N > 1 path on CPU: two processes (torch.distributed, gloo, 127.0.0.1) each build the rank's structure-only handle
and cross-check the ownership tables: the subtrees below the replicated camera root are dealt to exactly one rank
each, the deal is balanced, both ranks agree on it, and the sum over ranks of the per-rank partial root Hessians
(computed here with the CPU oracle's per-point cliques) equals the unsharded root assembly."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gtsam_personal_amd import LevenbergMarquardtOptimizer
        from gtsam_personal_amd.synthetic import make_bal
        graph, initial, _, ordering = make_bal(n_cam=20, n_pt=200, obs_per_point=5, seed=13)
        opt = LevenbergMarquardtOptimizer(graph, initial, ordering, device=-1, rank=rank, world_size=world)
        nf = opt.num_fronts()
        info = [opt.front_info(i) for i in range(nf)]
        owners = torch.tensor([f["owner"] for f in info], dtype=torch.int64)
        gathered = [torch.zeros_like(owners) for _ in range(world)]
        dist.all_gather(gathered, owners)
        for g in gathered:
            assert torch.equal(g, owners), "ranks disagree on the ownership table"
        root = nf - 1
        assert info[root]["cls"] == 1 and info[root]["owner"] == -1 and info[root]["n"] == 181
        leaves = [i for i in range(nf) if info[i]["parent"] == root]
        assert len(leaves) == 200
        counts = np.bincount([info[i]["owner"] for i in leaves], minlength=world)
        assert counts.sum() == 200 and counts.min() >= 200 // world - 15, counts
        # contiguous deal: owners are monotone over the children in front order
        ow = [info[i]["owner"] for i in leaves]
        assert ow == sorted(ow) or ow == sorted(ow, reverse=True)

        # reduction semantics: sum over ranks of the partial root assemblies == full assembly (oracle cliques)
        import oracle_harness as oh
        orc = oh.OracleProblem(graph, initial, ordering)
        orc.linearize()
        rc, _, _, _ = orc.solve(1e-3)
        assert rc == 0
        cl = orc.cliques()
        root_keys = cl[root][0]
        col = {}
        o = 0
        for k in root_keys:
            col[k] = o
            o += 9
        n = o + 1
        partial = np.zeros((n, n))
        full = np.zeros((n, n))
        for i in leaves:
            keys, nfk, rsd, parent = cl[i]
            S = rsd[:, 3:]  # [S d] of the point clique: its contribution to the root is -S^T S (+ the factor terms, same split)
            sep = keys[nfk:]
            idx = np.concatenate([np.arange(col[k], col[k] + 9) for k in sep] + [[n - 1]])
            U = -S.T @ S
            full[np.ix_(idx, idx)] += U
            if info[i]["owner"] == rank:
                partial[np.ix_(idx, idx)] += U
        t = torch.from_numpy(partial)
        dist.all_reduce(t)
        assert np.allclose(t.numpy(), full, rtol=1e-12, atol=1e-9)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, f"FAIL: {type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
