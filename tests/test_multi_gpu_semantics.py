"""N > 1 path on CPU: two processes over gloo.  Checks the host logic of
lightfm_amd/distributed.py (shard plan, per-rank seeds, the global merge schedule) and the merge
semantics the device code implements with RCCL (csrc/session.hip: merge_group) -- here with the
CPU oracle running each rank's segments and torch.distributed(gloo) carrying the all-reduces.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_plan_row_shards_balanced_and_complete():
    from lightfm_amd.distributed import plan_row_shards
    rng = np.random.RandomState(0)
    counts = rng.poisson(30, size=1000)
    counts[::50] += 2000  # heavy users
    for world in (1, 2, 3, 8):
        b = plan_row_shards(counts, world)
        assert b[0] == 0 and b[-1] == 1000 and np.all(np.diff(b) >= 0) and len(b) == world + 1
        per = [counts[b[r]:b[r + 1]].sum() for r in range(world)]
        assert sum(per) == counts.sum()
        # as equal as contiguous ranges allow: every boundary is within one row of its target
        assert all(abs(p - counts.sum() / world) <= 2 * counts.max() for p in per)
    assert len(plan_row_shards(np.zeros(5, int), 2)) == 3  # degenerate input: no crash


def test_local_shard_partitions_interactions():
    from lightfm_amd.distributed import local_shard
    from tests import helpers as H
    coo = H.make_interactions(200, 90, 4000, seed=1)
    parts = [local_shard(coo, r, 3)[0] for r in range(3)]
    assert sum(p.nnz for p in parts) == coo.nnz
    users = [set(p.row.tolist()) for p in parts]
    assert not (users[0] & users[1]) and not (users[1] & users[2]) and not (users[0] & users[2])
    assert all(p.shape == coo.shape for p in parts)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _merge(mode, start, local_list_or_none, local, allsum, world):
    """numpy restatement of csrc/session.hip merge_group for the item side (one rank's view)."""
    dW = {n: (local[n] - start[n]).astype(np.float32) for n in start}
    out = {}
    G_names = {"item_embeddings": "item_embedding_gradients", "item_biases": "item_bias_gradients"}
    if mode == "adagrad":
        tot = {g: allsum(dW[g]) for g in G_names.values()}
        for w, g in G_names.items():
            num = start[g] + 0.5 * dW[g]
            den = start[g] + 0.5 * tot[g]
            scale = np.where(den > num, np.sqrt(num / den), 1.0).astype(np.float32)
            out[w] = (start[w] + allsum((dW[w] * scale).astype(np.float32))).astype(np.float32)
            out[g] = (start[g] + tot[g]).astype(np.float32)
    else:
        for n in start:
            total = allsum(dW[n])
            if mode == "mean" and n in G_names:
                total = total / np.float32(world)
            out[n] = (start[n] + total).astype(np.float32)
    return out


def _dirty_union(start, local, allmax):
    """Rows whose W, G, b or bG differ from the interval's snapshot on ANY rank (csrc/session.hip:
    detect_dirty_kernel + the all-reduce MAX of the byte maps + compact_flagged_rows), ascending."""
    flags = np.zeros(len(start["item_biases"]), np.uint8)
    for n in start:
        diff = local[n] != start[n]
        flags |= (diff.any(axis=1) if diff.ndim == 2 else diff).astype(np.uint8)
    return np.flatnonzero(allmax(flags))


def _merge_sparse(mode, start, local, allsum, allmax, world):
    """numpy restatement of merge_group_sparse, one rank's view: (rows of the union, the merged deltas of those
    rows summed over ranks, this rank's own deltas) -- what complete_pending later applies."""
    rows = _dirty_union(start, local, allmax)
    sub_start = {n: start[n][rows] for n in start}
    sub_local = {n: local[n][rows] for n in start}
    merged = _merge(mode, sub_start, None, sub_local, allsum, world)
    total = {n: (merged[n] - sub_start[n]).astype(np.float32) for n in start}   # sum over ranks (rescaled)
    own = {n: (sub_local[n] - sub_start[n]).astype(np.float32) for n in start}  # unscaled
    return rows, total, own


ITEM_NAMES = ("item_embeddings", "item_embedding_gradients", "item_biases", "item_bias_gradients")
USER_NAMES = ("user_embeddings", "user_embedding_gradients", "user_biases", "user_bias_gradients")


def _policy():
    from lightfm_amd.distributed import MergePolicy
    return MergePolicy(merge_k=2, merge_min=150, merge_max=900, mode="adagrad")


def _rank_epochs(rank, world, coo, st, allsum, epochs=2, flavour="dense", allmax=None):
    """What DistributedFit.run does on one rank, with the CPU oracle as the epoch engine: segments of
    the rank's shuffled shard from the GLOBAL schedule, a merge of the item tables after each.
    flavour: "dense" (whole tables), "sparse" (the union of the rows that changed, applied at once) or
    "overlap" (the sparse merge applied at the NEXT merge / at the end of the epoch, while the rank has
    trained on: table += sum - own delta, snapshot += sum)."""
    from lightfm_amd.distributed import local_shard, merge_schedule, rank_seed, segment_positions
    from oracle import oracle
    from tests import helpers as H
    nu, ni = coo.shape
    shard, bounds = local_shard(coo, rank, world)
    item_f, user_f = H.identity_features(ni), H.identity_features(nu)
    rng = np.random.RandomState(rank_seed(11, rank))
    pol = _policy()
    history, merges = 0, 0
    pos_csr = H.positives_csr(shard)
    snap = {n: getattr(st, n).copy() for n in ITEM_NAMES}  # the tables at the start of the merge interval
    pending = None

    def land(exact):
        rows, total, own = pending
        for n in ITEM_NAMES:
            tab = getattr(st, n)
            new_snap = (snap[n][rows] + total[n]).astype(np.float32)
            tab[rows] = new_snap if exact else (tab[rows] + (total[n] - own[n])).astype(np.float32)
            snap[n][rows] = new_snap

    for _ in range(epochs):
        shuffle, seeds = H.epoch_inputs(shard, rng)
        pos = segment_positions(merge_schedule(history, coo.nnz, world, pol), shard.nnz)
        for j in range(len(pos) - 1):
            sub = np.ascontiguousarray(shuffle[pos[j]:pos[j + 1]])
            if len(sub):
                oracle.fit_warp(item_f, user_f, pos_csr, shard.row, shard.col, shard.data, shard.data, sub, st,
                                0.0, 0.0, seeds)
            local = {n: getattr(st, n) for n in ITEM_NAMES}
            if flavour == "dense":
                merged = _merge(pol.mode, snap, None, local, allsum, world)
                for n in ITEM_NAMES:
                    getattr(st, n)[...] = merged[n]
                    snap[n][...] = merged[n]
            else:
                if pending is not None:  # the previous exchange lands before this one's deltas are formed
                    land(False)
                pending = _merge_sparse(pol.mode, snap, local, allsum, allmax, world)
                if flavour == "sparse":
                    land(True)
                    pending = None
            merges += 1
        if pending is not None:  # lfm_session_comm_merge_flush at the end of the epoch
            land(False)
            pending = None
        history += coo.nnz
    return bounds, merges


def _worker(rank, world, port, out, flavour="dense"):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from tests import helpers as H
        nu, ni, d = 120, 80, 16
        coo = H.make_interactions(nu, ni, 3000, seed=5)
        st = oracle.State(ni, nu, d, np.random.RandomState(3))  # identical start on all ranks

        def allsum(x):
            t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32).copy())
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return t.numpy()

        def allmax(x):
            t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint8).copy())
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return t.numpy()

        bounds, merges = _rank_epochs(rank, world, coo, st, allsum, flavour=flavour, allmax=allmax)
        # the user rows of the other rank arrive once at the end (host plane): DistributedFit.gather_users
        for n in USER_NAMES:
            a = getattr(st, n)
            for r in range(world):
                b0, b1 = int(bounds[r]), int(bounds[r + 1])
                t = torch.from_numpy(np.ascontiguousarray(a[b0:b1]))
                dist.broadcast(t, src=r)
                a[b0:b1] = t.numpy()
        np.savez(out % rank, bounds=bounds, merges=merges, **{n: getattr(st, n) for n in ITEM_NAMES + USER_NAMES})
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("flavour", ["dense", "sparse", "overlap"])
def test_two_rank_gloo_schedule_matches_single_process_emulation(tmp_path, flavour):
    """World size 2 over gloo == the same algorithm emulated in one process: both replicas start
    from the same tables, train the segments of the global merge schedule on their shard, the item
    tables are merged after every segment (LFM_MERGE_ADAGRAD arithmetic), user rows are a disjoint
    union.  Both ranks must make the same number of collective calls (else this test hangs).
    flavour: the dense merge, the sparse merge over the union of changed rows (same result as the dense
    one, checked), and the overlapped sparse merge that lands one merge late."""
    import torch.multiprocessing as mp
    from lightfm_amd.distributed import local_shard
    from oracle import oracle
    from tests import helpers as H
    world, port = 2, _free_port()
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(world, port, out, flavour), nprocs=world, join=True)
    got = [np.load(out % r) for r in range(world)]
    assert int(got[0]["merges"]) == int(got[1]["merges"]) > 4

    # single-process emulation: the ranks advance segment by segment in lockstep
    import threading
    nu, ni, d = 120, 80, 16
    coo = H.make_interactions(nu, ni, 3000, seed=5)
    states = [oracle.State(ni, nu, d, np.random.RandomState(3)) for _ in range(world)]
    barrier = threading.Barrier(world)
    pending, results = {}, {}
    lock = threading.Lock()

    def make_collectives(rank):
        calls = [0]

        def collective(x, dtype, reduce):
            key = calls[0]
            calls[0] += 1
            with lock:
                pending.setdefault(key, []).append(np.asarray(x, dtype).copy())
            barrier.wait()
            with lock:
                if key not in results:
                    results[key] = reduce(pending[key])
            barrier.wait()
            return results[key].copy()

        def allsum(x):
            return collective(x, np.float32, lambda xs: np.sum(xs, axis=0, dtype=np.float32))

        def allmax(x):
            return collective(x, np.uint8, lambda xs: np.max(xs, axis=0))
        return allsum, allmax

    def run_rank(r):
        allsum, allmax = make_collectives(r)
        _rank_epochs(r, world, coo, states[r], allsum, flavour=flavour, allmax=allmax)

    threads = [threading.Thread(target=run_rank, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    bounds = local_shard(coo, 0, world)[1]
    for r in range(world):
        assert np.array_equal(got[r]["bounds"], bounds)
        for n in ITEM_NAMES:
            np.testing.assert_allclose(got[r][n], getattr(states[0], n), rtol=1e-6, atol=1e-7, err_msg=n)
            np.testing.assert_allclose(got[r][n], getattr(states[1], n), rtol=1e-6, atol=1e-7, err_msg=n)
        for n in USER_NAMES:
            for q in range(world):
                b0, b1 = int(bounds[q]), int(bounds[q + 1])
                np.testing.assert_allclose(got[r][n][b0:b1], getattr(states[q], n)[b0:b1], rtol=1e-6, atol=1e-7,
                                           err_msg=n)
    if flavour == "sparse":
        # the union of changed rows carries everything the dense merge does: same tables, bit for bit
        dense = [oracle.State(ni, nu, d, np.random.RandomState(3)) for _ in range(world)]
        pending.clear()
        results.clear()

        def run_dense(r):
            allsum, allmax = make_collectives(r)
            _rank_epochs(r, world, coo, dense[r], allsum, flavour="dense", allmax=allmax)
        threads = [threading.Thread(target=run_dense, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for n in ITEM_NAMES:
            np.testing.assert_array_equal(getattr(states[0], n), getattr(dense[0], n), err_msg=n)
    # every user row was trained by exactly one rank: the union changed rows of both shards
    fresh = oracle.State(ni, nu, d, np.random.RandomState(3)).user_embeddings
    moved = np.any(got[0]["user_embeddings"] != fresh, axis=1)
    assert moved[: bounds[1]].any() and moved[bounds[1]:].any()
