"""N > 1 path on CPU: two processes over gloo.  Checks the host logic of
lightfm_amd/distributed.py (shard plan, per-rank seeds) and the per-epoch merge semantics the
device code implements with RCCL (csrc/session.hip: merge_side) -- here with the CPU oracle
running each rank's epoch and torch.distributed(gloo) carrying the delta all-reduce.
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


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lightfm_amd.distributed import local_shard, merge_deltas, rank_seed
        from oracle import oracle
        from tests import helpers as H
        nu, ni, d = 120, 80, 16
        coo = H.make_interactions(nu, ni, 3000, seed=5)
        shard, bounds = local_shard(coo, rank, world)
        st = oracle.State(ni, nu, d, np.random.RandomState(3))  # identical start on all ranks
        item_f, user_f = H.identity_features(ni), H.identity_features(nu)
        rng = np.random.RandomState(rank_seed(11, rank))

        def allsum(x):
            t = torch.from_numpy(np.ascontiguousarray(x))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return t.numpy()

        item_names = ("item_embeddings", "item_embedding_gradients", "item_biases", "item_bias_gradients")
        user_names = ("user_embeddings", "user_embedding_gradients", "user_biases", "user_bias_gradients")
        user_start = {n: getattr(st, n).copy() for n in user_names}
        for _ in range(2):
            start = {n: getattr(st, n).copy() for n in item_names}
            shuffle, seeds = H.epoch_inputs(shard, rng)
            oracle.fit_warp(item_f, user_f, H.positives_csr(shard), shard.row, shard.col, shard.data,
                            shard.data, shuffle, st, 0.0, 0.0, seeds)
            for n in item_names:  # per-epoch merge of the replicated item side
                getattr(st, n)[...] = merge_deltas(start[n], getattr(st, n), allsum)
        for n in user_names:  # final union of the partitioned user side
            getattr(st, n)[...] = merge_deltas(user_start[n], getattr(st, n), allsum)
        np.savez(out % rank, bounds=bounds, **{n: getattr(st, n) for n in item_names + user_names})
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_merge_matches_single_process_emulation(tmp_path):
    """World size 2 over gloo == the same algorithm emulated in one process: both replicas start
    from the same tables, train their shard, item-side deltas are summed per epoch, user rows
    are a disjoint union."""
    import torch.multiprocessing as mp
    from lightfm_amd.distributed import local_shard, rank_seed
    from oracle import oracle
    from tests import helpers as H
    world, port = 2, _free_port()
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = [np.load(out % r) for r in range(world)]

    nu, ni, d = 120, 80, 16
    coo = H.make_interactions(nu, ni, 3000, seed=5)
    item_f, user_f = H.identity_features(ni), H.identity_features(nu)
    item_names = ("item_embeddings", "item_embedding_gradients", "item_biases", "item_bias_gradients")
    user_names = ("user_embeddings", "user_embedding_gradients", "user_biases", "user_bias_gradients")
    states = [oracle.State(ni, nu, d, np.random.RandomState(3)) for _ in range(world)]
    rngs = [np.random.RandomState(rank_seed(11, r)) for r in range(world)]
    shards = [local_shard(coo, r, world)[0] for r in range(world)]
    ustart = {n: getattr(states[0], n).copy() for n in user_names}
    for _ in range(2):
        start = {n: getattr(states[0], n).copy() for n in item_names}
        for r in range(world):
            shuffle, seeds = H.epoch_inputs(shards[r], rngs[r])
            oracle.fit_warp(item_f, user_f, H.positives_csr(shards[r]), shards[r].row, shards[r].col,
                            shards[r].data, shards[r].data, shuffle, states[r], 0.0, 0.0, seeds)
        for n in item_names:
            total = sum((getattr(s, n) - start[n]).astype(np.float32) for s in states)
            merged = (start[n] + total).astype(np.float32)
            for s in states:
                getattr(s, n)[...] = merged
    for n in user_names:
        total = sum((getattr(s, n) - ustart[n]).astype(np.float32) for s in states)
        for s in states:
            getattr(s, n)[...] = (ustart[n] + total).astype(np.float32)

    for r in range(world):
        assert np.array_equal(got[r]["bounds"], local_shard(coo, r, world)[1])
        for n in item_names + user_names:
            np.testing.assert_allclose(got[r][n], getattr(states[0], n), rtol=1e-6, atol=1e-7, err_msg=n)
    # every user row was trained by exactly one rank: the union changed rows of both shards
    b = got[0]["bounds"]
    moved = np.any(got[0]["user_embeddings"] != oracle.State(ni, nu, d, np.random.RandomState(3)).user_embeddings, axis=1)
    assert moved[: b[1]].any() and moved[b[1]:].any()
