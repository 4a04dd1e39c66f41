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


def _worker_subtrees(rank, world, port, q):
    """The distributed algorithm of the sharded solve, emulated with dense algebra on the CPU, one process per rank over gloo:
    ownership (whole camera subtrees, dense fronts included), the exchange (all-reduce of the partial assemblies of the replicated
    separator fronts, top level by level) and the replicated elimination of those fronts -- checked against the single-process
    oracle: the root's [R d] to 1e-9."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_harness as oh
        from gtsam_personal_amd import LevenbergMarquardtOptimizer
        from gtsam_personal_amd.graph import VAR_DIM
        from gtsam_personal_amd.synthetic import make_bal
        graph, initial, _, _ = make_bal(n_cam=64, n_pt=1200, obs_per_point=5, seed=3, window=8)
        ordering = oh.metis(graph)
        opt = LevenbergMarquardtOptimizer(graph, initial, ordering, device=-1, rank=rank, world_size=world)
        nf = opt.num_fronts()
        info = [opt.front_info(i) for i in range(nf)]
        owners = torch.tensor([f["owner"] for f in info], dtype=torch.int64)
        gathered = [torch.zeros_like(owners) for _ in range(world)]
        dist.all_gather(gathered, owners)
        for g in gathered:
            assert torch.equal(g, owners), "ranks disagree on the ownership table"
        keys_of = [opt.front(i, numeric=False)[0] for i in range(nf)]
        replicated = [i for i in range(nf) if info[i]["owner"] < 0]
        assert replicated and all(info[i]["cls"] == 1 for i in replicated)
        # dense fronts are dealt out too, every rank owns some, and a front's owner is its parent's unless the parent is replicated
        owned_dense = [info[i]["owner"] for i in range(nf) if info[i]["cls"] == 1 and info[i]["owner"] >= 0]
        assert set(owned_dense) == set(range(world)), owned_dense
        for i in range(nf):
            p = info[i]["parent"]
            if info[i]["owner"] >= 0 and p >= 0 and info[p]["owner"] >= 0:
                assert info[i]["owner"] == info[p]["owner"]
            if info[i]["owner"] < 0:
                assert p < 0 or info[p]["owner"] < 0  # replicated fronts form the top of the tree

        # ---- the numbers: whitened Jacobians of the oracle, lambda-damped normal equations
        lam = 1e-3
        orc = oh.OracleProblem(graph, initial, ordering)
        orc.linearize()
        rc, _, _, _ = orc.solve(lam)
        assert rc == 0
        cl = orc.cliques()
        types = {int(k): initial.type(k) for k in ordering}
        dim = {k: VAR_DIM[t] for k, t in types.items()}
        pos = {int(k): i for i, k in enumerate(ordering)}
        front_of_var = {}
        for i in range(nf):
            for k in keys_of[i][:info[i]["n_frontal_keys"]]:
                front_of_var[k] = i
        fkeys = graph.factor_keys_in_graph_order()
        facs_of = [[] for _ in range(nf)]
        for g, ks in enumerate(fkeys):
            first = min(ks, key=lambda k: pos[k])  # the factor sits in the front that eliminates its first variable
            facs_of[front_of_var[first]].append(g)
        children = [[] for _ in range(nf)]
        for i in range(nf):
            if info[i]["parent"] >= 0:
                children[info[i]["parent"]].append(i)

        def subtree(i):
            out, stack = [], [i]
            while stack:
                x = stack.pop()
                out.append(x)
                stack.extend(children[x])
            return out

        def layout(keys):
            off, o = {}, 0
            for k in keys:
                off[k] = o
                o += dim[k]
            return off, o

        def add_factors(H, off, n, fronts):
            for fi in fronts:
                for g in facs_of[fi]:
                    Ab = orc.jacobian(g)
                    idx = np.concatenate([np.arange(off[k], off[k] + dim[k]) for k in fkeys[g]] + [[n]])
                    H[np.ix_(idx, idx)] += Ab.T @ Ab

        def schur(H, ne):  # eliminate the first ne scalars
            A, B, C = H[:ne, :ne], H[:ne, ne:], H[ne:, ne:]
            return C - B.T @ np.linalg.solve(A, B)

        def owned_update(t):  # what the rank that owns subtree t hands to t's (replicated) parent: exact Schur complement
            fr = subtree(t)
            elim = [k for fi in sorted(fr) for k in keys_of[fi][:info[fi]["n_frontal_keys"]]]
            sep = keys_of[t][info[t]["n_frontal_keys"]:]
            off, n = layout(elim + sep)
            H = np.zeros((n + 1, n + 1))
            add_factors(H, off, n, fr)
            ne = sum(dim[k] for k in elim)
            H[np.arange(ne), np.arange(ne)] += lam  # the damping priors of the variables eliminated here
            return sep, schur(H, ne)

        updates = {}  # front -> (separator keys, update matrix incl. the rhs row / column), known on every rank once all-reduced
        for x in sorted(replicated):  # post-order: a replicated child comes before its replicated parent
            keys = keys_of[x]
            off, n = layout(keys)
            part = np.zeros((n + 1, n + 1))
            for c in children[x]:
                if info[c]["owner"] == rank:
                    sep, U = owned_update(c)
                elif info[c]["owner"] < 0 and rank == 0:
                    sep, U = updates[c]  # identical on every rank: enters the sum once
                else:
                    continue
                idx = np.concatenate([np.arange(off[k], off[k] + dim[k]) for k in sep] + [[n]])
                part[np.ix_(idx, idx)] += U
            if rank == 0:  # the front's own factors and damping: once
                add_factors(part, off, n, [x])
                nfs = sum(dim[k] for k in keys[:info[x]["n_frontal_keys"]])
                part[np.arange(nfs), np.arange(nfs)] += lam
            t = torch.from_numpy(part)
            dist.all_reduce(t)  # the exchange: the sum of the ranks' partial assemblies
            full = t.numpy()
            nfs = sum(dim[k] for k in keys[:info[x]["n_frontal_keys"]])
            updates[x] = (keys[info[x]["n_frontal_keys"]:], schur(full, nfs))
            # the replicated elimination against the single-process oracle: [R S d] of this clique
            R = np.linalg.cholesky(full[:nfs, :nfs]).T
            RSd = np.hstack([R, np.linalg.solve(R.T, full[:nfs, nfs:])])
            ck, cnfk, crsd, _ = cl[x]
            assert ck == keys and crsd.shape == RSd.shape
            assert np.allclose(RSd, crsd, rtol=1e-9, atol=1e-9 * np.abs(crsd).max()), (x, np.abs(RSd - crsd).max())
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"FAIL: {type(e).__name__}: {e}\n{traceback.format_exc()}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_camera_subtree_sharding_gloo(world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_harness as oh
    if not oh.have_ref():
        pytest.skip("oracle/_ref (the reference's METIS) not built")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_subtrees, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
