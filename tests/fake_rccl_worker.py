"""One rank of tests/test_fake_rccl_multirank.py: K PROCESSES on the ONE GPU of the box run the product's multi-GPU
driver (lightfm_amd.distributed.DistributedFit: replicated item tables merged through ncclAllReduce) over
tests/fake_rccl.hip -- real RCCL refuses two ranks on one device, the stand-in (selected with LIGHTFM_AMD_RCCL, the
loader's own override) does not.  torch.distributed (gloo) is the rendezvous, as in a real job.

    LIGHTFM_AMD_RCCL=tests/_bin/libfake_rccl.so RANK=r WORLD_SIZE=K MASTER_ADDR=127.0.0.1 MASTER_PORT=p \
        python tests/fake_rccl_worker.py

A. DETERMINISTIC runs (one interaction per launch: the Hogwild kernels are then sequential), every merge flavour x mode:
   the rank's tables after 2 epochs must equal, BIT FOR BIT, what the same schedule gives with K sessions in ONE process
   merged by lfm_sessions_merge_local / _local_sparse / _local_hot (the arithmetic the one-GPU emulations of DESIGN.md
   were measured with) -- identity and hybrid models, hot rows, shared user features.
B. FULL-CONCURRENCY runs of the shipped defaults: every rank ends with the same item tables (compared across ranks
   over the rendezvous), finite, the same number of merges and bytes, and a model that ranks its positives.
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ITEM = ("item_embeddings", "item_embedding_gradients", "item_biases", "item_bias_gradients")
USER = ("user_embeddings", "user_embedding_gradients", "user_biases", "user_bias_gradients")


def _models(world, seed, loss, d, **kw):
    """The K ranks' models as the ranks build them: own RandomState streams, rank 0's initial tables."""
    from lightfm_amd import LightFM
    from lightfm_amd.distributed import rank_seed
    return [LightFM(no_components=d, loss=loss, random_state=rank_seed(seed, r), **kw) for r in range(world)]


def emulate(world, coo, item_features, user_features, policy, epochs, seed, loss, d):
    """What K ranks of DistributedFit compute, in ONE process: K sessions with their shards, the driver's own schedule
    (merge_plan / segment_positions), the local merges.  Returns the K trained models (item tables identical)."""
    from lightfm_amd._lightfm_fast import CSRMatrix, FastLightFM, make_opts
    from lightfm_amd.distributed import hot_rows, local_shard, merge_plan, segment_positions
    from lightfm_amd.lightfm import _Session, _WEIGHTS
    n_users, n_items = coo.shape
    models = _models(world, seed, loss, d)
    shared_users = user_features is not None
    sessions, structs, shards, ranges = [], [], [], []
    bounds = None
    for r, m in enumerate(models):
        user_f, item_f = m._construct_feature_matrices(n_users, n_items, user_features, item_features)
        m._initialize(d, item_f.shape[1], user_f.shape[1])
        if r > 0:  # DistributedFit broadcasts rank 0's embeddings
            m.item_embeddings[...] = models[0].item_embeddings
            m.user_embeddings[...] = models[0].user_embeddings
        shard, bounds = local_shard(coo, r, world, bounds=bounds, rebase=True)
        b0, b1 = int(bounds[r]), int(bounds[r + 1])
        arrays = [getattr(m, name)[b0:b1] if (name.startswith("user") and not shared_users) else getattr(m, name)
                  for name in _WEIGHTS]
        st = FastLightFM(*arrays, d, 0, m.learning_rate, m.rho, m.epsilon, m.max_sampled)
        if shared_users:
            ruf = user_f[b0:b1].tocsr()
            ruf.sort_indices()
        else:
            ruf = sp.identity(b1 - b0, dtype=np.float32, format="csr")
        s = _Session(st, CSRMatrix(item_f), CSRMatrix(ruf), device=0)
        s.set_interactions(None, np.ascontiguousarray(shard.row), np.ascontiguousarray(shard.col), shard.data, shard.data)
        s.build_positives(b1 - b0, n_items)
        sessions.append(s)
        structs.append(st)
        shards.append(shard)
        ranges.append((b0, b1))
    sides = 1 | (2 if shared_users else 0)
    hot = [hot_rows(item_features, policy.hot_share, 2.0), hot_rows(user_features if shared_users else None, policy.hot_share, 1.0)]
    has_hot = len(hot[0]) > 0 or len(hot[1]) > 0
    for s in sessions:
        s.merge_begin(sides)
        for side in (0, 1):
            if len(hot[side]):
                s.set_hot_rows(side, hot[side])
    n_rep = max(models[0].item_embeddings.shape[0], models[0].user_embeddings.shape[0] if shared_users else 0)
    history = 0
    try:
        for _ in range(epochs):
            seeds = []
            for r, (m, s) in enumerate(zip(models, sessions)):  # the draws of DistributedFit.epoch, in its order
                keys = m.random_state.randint(0, np.iinfo(np.int32).max, size=624)
                s.device_shuffle(int(keys[0]), int(keys[1]))
                seeds.append(np.ascontiguousarray(m.random_state.randint(0, np.iinfo(np.int32).max, size=1).astype(np.uint32)))
            fr, kinds = merge_plan(history, coo.nnz, world, policy, n_rep, has_hot)
            pos = [segment_positions(fr, sh.nnz) for sh in shards]
            for j in range(len(fr) - 1):
                for r, s in enumerate(sessions):
                    n = shards[r].nnz
                    opts, _ = make_opts()
                    opts.history = (history + int(round(coo.nnz * pos[r][j] / max(1, n)))) // world
                    opts.pos_begin, opts.pos_end = int(pos[r][j]), int(pos[r][j + 1])
                    if pos[r][j + 1] > pos[r][j]:
                        s.epoch(loss, 0.0, 0.0, 5, 10, seeds[r], opts)
                if not policy.sparse:
                    if kinds[j] == "full":
                        _Session.merge_local(sessions, sides, policy.mode_id())
                elif kinds[j] == "hot":
                    _Session.merge_local_hot(sessions, sides, policy.mode_id(), overlap=policy.overlap)
                else:
                    _Session.merge_local_sparse(sessions, sides, policy.mode_id(), overlap=policy.overlap)
            if policy.sparse:
                _Session.merge_local_flush(sessions)
            history += coo.nnz
        for s, st in zip(sessions, structs):
            s.sync_to_host(st)
    finally:
        for s in sessions:
            s.close()
    return models, ranges


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    from lightfm_amd import LightFM, _native as N
    assert N.device_count() > 0, "no HIP device"
    assert os.environ.get("LIGHTFM_AMD_RCCL", "").endswith("libfake_rccl.so"), "the test must select the stand-in"
    N.check(N.lib().lfm_comm_preload())  # the library named by LIGHTFM_AMD_RCCL, resolved before torch is imported
    import torch
    import torch.distributed as dist
    from lightfm_amd import synthetic
    from lightfm_amd.distributed import DistributedFit, MergePolicy, rank_seed
    from lightfm_amd.options import options
    from tests import helpers as H
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    def agree(arrays, what, exact=True):
        """Every rank holds the same item tables: the same BITS after synchronous merges (table := snapshot := snapshot +
        sum on every rank); after OVERLAPPED merges the snapshots are bit-identical but a rank's table is
        fl(table + fl(sum - own delta)) -- the same value up to the rounding of that one addition."""
        t = torch.from_numpy(np.concatenate([np.ascontiguousarray(a, dtype=np.float32).ravel() for a in arrays]))
        every = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        for r, other in enumerate(every):
            if exact:
                assert torch.equal(other.view(torch.int32), every[0].view(torch.int32)), "%s: rank %d differs from rank 0" % (what, r)
            else:
                np.testing.assert_allclose(other.numpy(), every[0].numpy(), rtol=2e-5, atol=1e-6, err_msg="%s: rank %d" % (what, r))

    # ---- A: deterministic, against the one-process emulation, bit for bit
    nu, ni, d = 400, 300, 16
    coo = H.make_interactions(nu, ni, 6000, seed=23, zipf=0.7)
    tags = H.tag_features(ni, 12, 3, seed=5)            # [identity | 3 of 12 tags]: hot rows
    utags = H.tag_features(nu, 20, 2, seed=6)           # shared user features: user tables merged too (sides = 3)
    cases = [
        ("dense-adagrad", "warp", None, None, dict(mode="adagrad", sparse=False)),
        ("sparse-sum", "warp", None, None, dict(mode="sum")),
        ("sparse-adagrad", "warp", None, None, dict(mode="adagrad")),
        ("overlap-adagrad", "warp", None, None, dict(mode="adagrad", overlap=True)),
        ("hot-bpr-tags", "bpr", tags, None, dict(mode="adagrad", hot_max=700)),
        ("shared-users-mean", "warp", tags, utags, dict(mode="mean", hot_max=900)),
    ]
    for name, loss, itf, usf, pol_kw in cases:
        options.set(mode="parallel", launches_per_epoch=1 << 20, ramp_k=0, max_waves=0, debug=0, shuffle_ahead=False)
        policy = MergePolicy(merge_k=2, merge_min=800, merge_max=2500, **pol_kw)
        model = LightFM(no_components=d, loss=loss, random_state=rank_seed(77, rank))
        fit = DistributedFit(model, coo, rank, world, device=0, dist=dist, policy=policy, item_features=itf, user_features=usf)
        start = model.item_embeddings.copy()
        fit.run(2)
        merges, nbytes = fit.merges, fit.merge_bytes
        fit.close()
        assert merges >= 6, (name, merges)
        if "hot" in name or "shared" in name:
            assert fit.has_hot, name
        want, ranges = emulate(world, coo, itf, usf, policy, 2, 77, loss, d)
        b0, b1 = ranges[rank]
        for n_ in ITEM:
            a, b = getattr(model, n_), getattr(want[rank], n_)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "%s rank %d: %s differs from the one-process emulation at %d cells (max abs %g)" % (
                name, rank, n_, int((a != b).sum()), float(np.abs(a - b).max()))
        for n_ in USER:
            a, b = getattr(model, n_), getattr(want[rank], n_)
            if usf is None:
                a, b = a[b0:b1], b[b0:b1]
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "%s rank %d: %s differs" % (name, rank, n_)
        assert not np.array_equal(start, model.item_embeddings), name + ": nothing was trained"
        agree([getattr(model, n_) for n_ in ITEM], name, exact=not policy.overlap)
        t = torch.tensor([merges, nbytes], dtype=torch.int64)
        every = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        assert all(torch.equal(o, every[0]) for o in every), (name, [o.tolist() for o in every])
        if rank == 0:
            print("FAKE_RCCL case %-20s ok: %d merges, %d bytes to the all-reduce per rank" % (name, merges, nbytes), flush=True)
        dist.barrier()

    # ---- B: the shipped defaults at full concurrency
    nu, ni, d = 6000, 4000, 64
    coo = synthetic.make_interactions(nu, ni, 400_000, seed=3)
    tags_b = synthetic.tag_item_features(ni, n_tags=200, per_item=4)
    for name, loss, itf, pol_kw in (("c2-like", "warp", None, dict()),
                                    ("c2-like-overlap", "warp", None, dict(overlap=True)),
                                    ("c3-like-hot", "bpr", tags_b, dict(hot_max=1 << 15))):
        options.set(mode="parallel", launches_per_epoch=0, ramp_k=0, max_waves=0, debug=0, shuffle_ahead=False)
        policy = MergePolicy(merge_min=4096, merge_max=1 << 17, **pol_kw)
        model = LightFM(no_components=d if itf is None else 32, loss=loss, random_state=rank_seed(5, rank))
        fit = DistributedFit(model, coo, rank, world, device=0, dist=dist, policy=policy, item_features=itf)
        start = model.item_embeddings.copy()
        stats = fit.run(3)
        assert all(int(o.kernel_used) in (1, 2) for o in stats)
        fit.gather_users()
        merges = fit.merges
        fit.close()
        assert merges >= 8, (name, merges)
        for n_ in ITEM + USER:
            assert np.isfinite(getattr(model, n_)).all(), (name, n_)
        assert not np.array_equal(start, model.item_embeddings)
        agree([getattr(model, n_) for n_ in ITEM], name, exact=not policy.overlap)
        rows, cols = np.ascontiguousarray(coo.row), np.ascontiguousarray(coo.col)
        neg = np.random.RandomState(1).randint(0, ni, size=coo.nnz).astype(np.int32)
        acc = float(np.mean(model.predict(rows, cols, item_features=itf) > model.predict(rows, neg, item_features=itf)))
        assert acc > 0.8, (name, acc)
        if rank == 0:
            print("FAKE_RCCL case %-20s ok: %d merges, pairwise accuracy %.3f" % (name, merges, acc), flush=True)
        dist.barrier()
    dist.destroy_process_group()
    print("FAKE_RCCL_WORKER_OK rank %d of %d" % (rank, world), flush=True)


if __name__ == "__main__":
    main()
