"""Owner-sharded item tables (include/lfm_hip.h: lfm_sessions_share_items_local; device.hpp: ItemShards) -- the
one-device form of the multi-GPU decomposition for item sides too large to replicate and merge (BASELINE config C4).

K sessions of one device, each with a contiguous range of the users and ITS OWN copy of the item tables in which
every row it does not own is poisoned (1e30): a kernel that read or wrote a row anywhere but at its owner would
show in the samples, in the poison, or in both.

  * frozen weights (sample_weight = 0): every session's (negative, sampled) per position and its four counters equal
    the CPU oracle's on the true, unsharded model -- the sharded addressing reads exactly the reference's rows;
  * training: the K sessions train for two epochs side by side; afterwards no poisoned (non-owned) row of any
    session has changed, the owners' rows have, and the model assembled from the owners ranks the training positives
    above random items like a one-session fit of the same data does.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu

POISON = np.float32(1e30)
K = 4


def _setup(frozen, nu=2000, ni=3001, nnz=90000):
    from lightfm_amd import _native
    from lightfm_amd._lightfm_fast import CSRMatrix, FastLightFM
    from lightfm_amd.distributed import local_shard
    from lightfm_amd.lightfm import _Session
    assert _native.device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    d = 64  # 3001 items over 4 owners: 751, 751, 751, 748 rows
    coo = H.make_interactions(nu, ni, nnz, seed=31, zipf=0.7)
    rng = np.random.RandomState(13)
    st = oracle.State(ni, nu, d, rng, max_sampled=10)
    a = 3.0 / d ** 0.25
    st.item_embeddings *= 2 * d * a
    st.user_embeddings *= 2 * d * a
    st.item_biases[:] = rng.randn(ni).astype(np.float32) * 0.3
    st.user_biases[:] = rng.randn(nu).astype(np.float32) * 0.3
    rps = (ni + K - 1) // K
    sessions, parts = [], []
    for j in range(K):
        shard, bounds = local_shard(coo, j, K, rebase=True)
        b0, b1 = int(bounds[j]), int(bounds[j + 1])
        mine = st.copy()
        own = np.zeros(ni, bool)
        own[j * rps:min(ni, (j + 1) * rps)] = True
        for name in oracle.ARRAYS:
            arr = getattr(mine, name)
            if name.startswith("item") and "momentum" not in name:
                arr[~own] = POISON
            if name.startswith("user"):
                setattr(mine, name, np.ascontiguousarray(arr[b0:b1]))
        fl = FastLightFM(*mine.arrays(), d, 0, mine.lr, mine.rho, mine.eps, mine.max_sampled)
        s = _Session(fl, CSRMatrix(H.identity_features(ni)), CSRMatrix(H.identity_features(b1 - b0)))
        w = np.zeros_like(shard.data) if frozen else shard.data
        s.set_interactions(None, np.ascontiguousarray(shard.row), np.ascontiguousarray(shard.col), shard.data, w)
        s.build_positives(b1 - b0, ni)
        sessions.append(s)
        parts.append(dict(shard=shard, range=(b0, b1), own=own, state=mine, struct=fl, weight=w))
    _Session.share_items_local(sessions)
    return coo, st, sessions, parts, (nu, ni, d)


def _close(sessions):
    for s in sessions:
        s.close()


def test_frozen_weights_samples_exact_over_sharded_item_tables():
    from lightfm_amd._lightfm_fast import make_opts
    from lightfm_amd.options import options
    coo, st, sessions, parts, (nu, ni, d) = _setup(frozen=True)
    options.set(mode="parallel", ramp_k=-1, launches_per_epoch=3, debug=0)
    try:
        for j, (s, pt) in enumerate(zip(sessions, parts)):
            shard = pt["shard"]
            n = shard.nnz
            rng = np.random.RandomState(100 + j)
            shuffle = np.arange(n, dtype=np.int32)
            rng.shuffle(shuffle)
            seeds = rng.randint(0, np.iinfo(np.int32).max, size=1).astype(np.uint32)
            s.upload_shuffle(shuffle)
            opts, logs = make_opts(n, want_log=True)
            s.epoch("warp", 0.0, 0.0, 5, 10, seeds, opts)
            assert opts.kernel_used == 1 and opts.tile_ng == 4
            # the oracle on the TRUE model (no poison), this rank's users
            b0, b1 = pt["range"]
            ref = st.copy()
            for name in oracle.ARRAYS:
                if name.startswith("user"):
                    setattr(ref, name, np.ascontiguousarray(getattr(st, name)[b0:b1]))
            o = oracle.Opts(n, rng_mode=1, log=True)
            oracle.fit_warp(H.identity_features(ni), H.identity_features(b1 - b0), H.positives_csr(shard), shard.row, shard.col,
                            shard.data, pt["weight"], shuffle, ref, 0.0, 0.0, seeds, o)
            neg, sampled = logs
            assert np.array_equal(sampled, o.sampled), "session %d: sample counts differ" % j
            assert np.array_equal(neg, o.neg), "session %d: negatives differ" % j
            assert list(opts.counters) == o.counters
            assert o.counters[2] > n // 4
    finally:
        _close(sessions)


def test_sharded_updates_match_the_unsharded_oracle():
    """NON-ZERO updates through the SHARDED instantiation of the steady-state tile kernel: the K sessions run one
    after the other, one interaction per launch (sequential), two epochs; the model assembled from the owners' rows
    and the sessions' user rows equals the oracle's on the TRUE unsharded model run over the same shards in the same
    order -- within one float32 ulp (publication is old + float32(new - old)), bit-identical where that is exact."""
    from lightfm_amd._lightfm_fast import make_opts
    from lightfm_amd.options import options
    coo, st, sessions, parts, (nu, ni, d) = _setup(frozen=False, nu=240, ni=3001, nnz=1500)
    ref = st.copy()
    try:
        rng = np.random.RandomState(77)
        for epoch in range(2):
            for j, (s, pt) in enumerate(zip(sessions, parts)):
                shard = pt["shard"]
                n = shard.nnz
                shuffle = np.arange(n, dtype=np.int32)
                rng.shuffle(shuffle)
                seeds = rng.randint(0, np.iinfo(np.int32).max, size=1).astype(np.uint32)
                s.upload_shuffle(shuffle)
                options.set(mode="parallel", launches_per_epoch=n, update_mode=0, debug=0)
                opts, logs = make_opts(n, want_log=True)
                s.epoch("warp", 0.0, 0.0, 5, 10, seeds, opts)
                assert opts.kernel_used == 1 and opts.tile_ng == 4 and opts.tile_ahead == 1 and opts.launches == n
                # the oracle: the same shard against the global item tables and this session's slice of the user tables
                b0, b1 = pt["range"]
                view = object.__new__(oracle.State)
                view.__dict__.update(ref.__dict__)
                for name in oracle.ARRAYS:
                    if name.startswith("user"):
                        setattr(view, name, getattr(ref, name)[b0:b1])
                o = oracle.Opts(n, rng_mode=1, log=True)
                oracle.fit_warp(H.identity_features(ni), H.identity_features(b1 - b0), H.positives_csr(shard), shard.row,
                                shard.col, shard.data, shard.data, shuffle, view, 0.0, 0.0, seeds, o)
                neg, sampled = logs
                assert np.array_equal(sampled, o.sampled), "epoch %d session %d: sample counts differ" % (epoch, j)
                assert np.array_equal(neg, o.neg), "epoch %d session %d: negatives differ" % (epoch, j)
                assert list(opts.counters) == o.counters
        final = st.copy()
        for j, (s, pt) in enumerate(zip(sessions, parts)):
            s.sync_to_host(pt["struct"])
            mine, own = pt["state"], pt["own"]
            b0, b1 = pt["range"]
            for name in ("item_embeddings", "item_embedding_gradients", "item_biases", "item_bias_gradients"):
                arr = getattr(mine, name)
                assert np.all(arr[~own] == POISON), "session %d wrote %s rows it does not own" % (j, name)
                getattr(final, name)[own] = arr[own]
            for name in oracle.ARRAYS:
                if name.startswith("user"):
                    getattr(final, name)[b0:b1] = getattr(mine, name)
    finally:
        _close(sessions)
    assert not np.array_equal(final.item_embeddings, st.item_embeddings)
    assert not np.array_equal(final.item_biases, st.item_biases)
    H.assert_states_within_ulps(final, ref, ulps=4)


def test_training_writes_only_the_owners_rows():
    from lightfm_amd._lightfm_fast import make_opts
    from lightfm_amd.options import options
    coo, st, sessions, parts, (nu, ni, d) = _setup(frozen=False)
    options.set(mode="parallel", debug=0)
    try:
        rng = np.random.RandomState(5)
        history = 0
        for epoch in range(2):
            for s, pt in zip(sessions, parts):
                keys = rng.randint(0, np.iinfo(np.int32).max, size=3)
                s.device_shuffle(int(keys[0]), int(keys[1]))
                opts, _ = make_opts()
                opts.history = history
                s.epoch("warp", 0.0, 0.0, 5, 10, np.array([keys[2]], np.uint32), opts)
                history += pt["shard"].nnz // K
        final = st.copy()
        for j, (s, pt) in enumerate(zip(sessions, parts)):
            s.sync_to_host(pt["struct"])
            mine, own = pt["state"], pt["own"]
            b0, b1 = pt["range"]
            for name in ("item_embeddings", "item_embedding_gradients", "item_biases", "item_bias_gradients"):
                arr = getattr(mine, name)
                assert np.all(arr[~own] == POISON), "session %d wrote %s rows it does not own" % (j, name)
                assert np.isfinite(arr[own]).all() and np.abs(arr[own]).max() < 1e6
                getattr(final, name)[own] = arr[own]
            for name in oracle.ARRAYS:
                if name.startswith("user"):
                    getattr(final, name)[b0:b1] = getattr(mine, name)
        changed = np.any(final.item_embeddings != st.item_embeddings, axis=1)
        assert changed.mean() > 0.5, "the owners' rows must have been trained by ALL sessions"
        assert np.all(final.item_embedding_gradients >= 1.0)
    finally:
        _close(sessions)
    item_f, user_f = H.identity_features(ni), H.identity_features(nu)
    r = np.random.RandomState(0)
    pos = oracle.predict(item_f, user_f, coo.row, coo.col, final)
    neg = oracle.predict(item_f, user_f, coo.row, r.randint(0, ni, size=coo.nnz).astype(np.int32), final)
    pos0 = oracle.predict(item_f, user_f, coo.row, coo.col, st)
    neg0 = oracle.predict(item_f, user_f, coo.row, r.randint(0, ni, size=coo.nnz).astype(np.int32), st)
    assert np.mean(pos > neg) > np.mean(pos0 > neg0) + 0.15, (np.mean(pos > neg), np.mean(pos0 > neg0))


def test_sharded_sessions_refuse_kernels_that_cannot_address_them():
    from lightfm_amd._lightfm_fast import make_opts
    from lightfm_amd.options import options
    coo, st, sessions, parts, _ = _setup(frozen=True)
    try:
        options.set(mode="parallel", warp_kernel=1)  # "do not use the tile kernel"
        s, pt = sessions[0], parts[0]
        s.upload_shuffle(np.arange(pt["shard"].nnz, dtype=np.int32))
        opts, _ = make_opts()
        with pytest.raises(NotImplementedError):
            s.epoch("warp", 0.0, 0.0, 5, 10, np.array([1], np.uint32), opts)
    finally:
        _close(sessions)
