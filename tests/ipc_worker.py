"""One rank of tests/test_sharded_items_ipc.py: K PROCESSES on ONE GPU share owner-sharded item tables through HIP IPC
(include/lfm_hip.h: lfm_session_export_items / lfm_session_share_items_ipc / lfm_session_gather_shared_items).
torch.distributed (gloo) is the rendezvous only: it hands the exports round and carries the barriers.

    RANK=r WORLD_SIZE=K MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/ipc_worker.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

POISON = np.float32(1e30)
ITEM_TABLES = ("item_embeddings", "item_embedding_gradients", "item_biases", "item_bias_gradients")


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    from lightfm_amd import LightFM, _native as N
    assert N.device_count() > 0, "no HIP device"   # (our HIP runtime is in the process before torch's)
    import torch
    import torch.distributed as dist
    from lightfm_amd._lightfm_fast import CSRMatrix, FastLightFM, make_opts
    from lightfm_amd.distributed import DistributedFit, local_shard
    from lightfm_amd.lightfm import _Session
    from lightfm_amd.options import options
    from oracle import oracle
    from tests import helpers as H
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    nu, ni, d = 1500, 3001, 64
    coo = H.make_interactions(nu, ni, 60000, seed=31, zipf=0.7)
    rng = np.random.RandomState(13)
    st = oracle.State(ni, nu, d, rng, max_sampled=10)
    a = 3.0 / d ** 0.25
    st.item_embeddings *= 2 * d * a
    st.user_embeddings *= 2 * d * a
    st.item_biases[:] = rng.randn(ni).astype(np.float32) * 0.3
    st.user_biases[:] = rng.randn(nu).astype(np.float32) * 0.3
    rps = (ni + world - 1) // world
    shard, bounds = local_shard(coo, rank, world, rebase=True)
    b0, b1 = int(bounds[rank]), int(bounds[rank + 1])
    own = np.zeros(ni, bool)
    own[rank * rps:min(ni, (rank + 1) * rps)] = True
    mine = st.copy()
    for name in oracle.ARRAYS:
        arr = getattr(mine, name)
        if name in ITEM_TABLES:
            arr[~own] = POISON   # a kernel that read or wrote a row anywhere but at its owner would show
        if name.startswith("user"):
            setattr(mine, name, np.ascontiguousarray(arr[b0:b1]))
    fl = FastLightFM(*mine.arrays(), d, 0, mine.lr, mine.rho, mine.eps, mine.max_sampled)
    s = _Session(fl, CSRMatrix(H.identity_features(ni)), CSRMatrix(H.identity_features(b1 - b0)), device=0)
    rows, cols = np.ascontiguousarray(shard.row), np.ascontiguousarray(shard.col)
    n = shard.nnz

    blob = torch.frombuffer(bytearray(s.export_items()), dtype=torch.uint8).clone()
    blobs = [torch.empty_like(blob) for _ in range(world)]
    dist.all_gather(blobs, blob)
    s.share_items_ipc([bytes(b.numpy().tobytes()) for b in blobs], rank)
    dist.barrier()

    # ---- A: frozen weights -- every position's (negative, sampled) and the counters equal the oracle's on the TRUE model
    zeros = np.zeros_like(shard.data)
    s.set_interactions(None, rows, cols, shard.data, zeros)
    s.build_positives(b1 - b0, ni)
    r = np.random.RandomState(100 + rank)
    shuffle = np.arange(n, dtype=np.int32)
    r.shuffle(shuffle)
    seeds = r.randint(0, np.iinfo(np.int32).max, size=1).astype(np.uint32)
    s.upload_shuffle(shuffle)
    options.set(mode="parallel", ramp_k=-1, launches_per_epoch=3, debug=0)
    opts, logs = make_opts(n, want_log=True)
    s.epoch("warp", 0.0, 0.0, 5, 10, seeds, opts)
    assert opts.kernel_used == 1 and opts.tile_ng == 4 and opts.tile_ahead == 1
    ref = st.copy()
    for name in oracle.ARRAYS:
        if name.startswith("user"):
            setattr(ref, name, np.ascontiguousarray(getattr(st, name)[b0:b1]))
    o = oracle.Opts(n, rng_mode=1, log=True)
    oracle.fit_warp(H.identity_features(ni), H.identity_features(b1 - b0), H.positives_csr(shard), shard.row, shard.col,
                    shard.data, zeros, shuffle, ref, 0.0, 0.0, seeds, o)
    assert np.array_equal(logs[1], o.sampled), "rank %d: sample counts differ" % rank
    assert np.array_equal(logs[0], o.neg), "rank %d: negatives differ" % rank
    assert list(opts.counters) == o.counters and o.counters[2] > n // 4
    dist.barrier()

    # ---- B: all ranks train at once against the owners' rows
    s.set_interactions(None, rows, cols, shard.data, shard.data)
    s.build_positives(b1 - b0, ni)
    options.set(mode="parallel", ramp_k=0, launches_per_epoch=0, debug=0)
    history = 0
    for e in range(2):
        s.device_shuffle(1000 + 10 * e + rank, 7)
        opts, _ = make_opts()
        opts.history = history
        s.epoch("warp", 0.0, 0.0, 5, 10, np.array([50 + 10 * e + rank], np.uint32), opts)
        assert opts.tile_ahead == 1
        history += n
    assert s.check_finite()
    dist.barrier()
    s.sync_to_host(fl)
    for name in ITEM_TABLES:
        arr = getattr(mine, name)
        assert np.all(arr[~own] == POISON), "rank %d wrote %s rows it does not own" % (rank, name)
        assert np.isfinite(arr[own]).all() and np.abs(arr[own]).max() < 1e6
    changed = np.any(mine.item_embeddings[own] != st.item_embeddings[own], axis=1)
    assert changed.mean() > 0.5, "rank %d's rows must have been trained by ALL ranks (%.2f changed)" % (rank, changed.mean())
    dist.barrier()
    s.gather_shared_items()
    dist.barrier()
    s.sync_to_host(fl)
    sums = []
    for name in ITEM_TABLES:
        arr = getattr(mine, name)
        assert not np.any(arr == POISON) and np.isfinite(arr).all(), "rank %d: %s not fully gathered" % (rank, name)
        sums.append(float(arr.astype(np.float64).sum()))
        sums.append(float(np.abs(arr.astype(np.float64)).sum()))
    t = torch.tensor(sums, dtype=torch.float64)
    every = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    for other in every:
        assert torch.equal(other, every[0]), "the ranks hold different item tables after the gather"
    # the gathered model ranks this rank's training positives above random items far better than the start did
    view, view0 = mine, st.copy()
    for name in oracle.ARRAYS:
        if name.startswith("user"):
            setattr(view0, name, np.ascontiguousarray(getattr(st, name)[b0:b1]))
    item_f, user_f = H.identity_features(ni), H.identity_features(b1 - b0)
    rr = np.random.RandomState(0)
    negs = rr.randint(0, ni, size=n).astype(np.int32)
    acc = np.mean(oracle.predict(item_f, user_f, shard.row, shard.col, view) > oracle.predict(item_f, user_f, shard.row, negs, view))
    acc0 = np.mean(oracle.predict(item_f, user_f, shard.row, shard.col, view0) > oracle.predict(item_f, user_f, shard.row, negs, view0))
    assert acc > acc0 + 0.15, (acc, acc0)
    dist.barrier()
    s.close()

    # ---- C: the same through the product's driver
    options.set(mode="parallel", ramp_k=0, launches_per_epoch=0, debug=0)
    model = LightFM(no_components=32, loss="warp", random_state=5)
    fit = DistributedFit(model, coo, rank, world, device=0, dist=dist, item_tables="owner")
    before = model.item_embeddings.copy()
    stats = fit.run(3)
    assert all(int(o_.tile_ahead) == 1 and int(o_.kernel_used) == 1 for o_ in stats)
    assert fit.merges == 0
    t = torch.tensor([float(model.item_embeddings.astype(np.float64).sum()), float(np.abs(model.item_biases).sum())],
                     dtype=torch.float64)
    every = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    for other in every:
        assert torch.equal(other, every[0])
    assert np.isfinite(model.item_embeddings).all() and not np.array_equal(before, model.item_embeddings)
    fit.gather_users()
    pos = model.predict(np.ascontiguousarray(coo.row), np.ascontiguousarray(coo.col))
    neg = model.predict(np.ascontiguousarray(coo.row), np.random.RandomState(1).randint(0, ni, size=coo.nnz).astype(np.int32))
    assert np.mean(pos > neg) > 0.75, np.mean(pos > neg)
    fit.close()
    dist.barrier()
    dist.destroy_process_group()
    print("IPC_WORKER_OK rank %d of %d" % (rank, world), flush=True)


if __name__ == "__main__":
    main()
