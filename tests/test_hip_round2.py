"""GPU tests of the round-2 surface: adadelta under full concurrency, parallel-mode L2
regularisation, epoch segments, the device-built positives lookup, the merge arithmetic of the
multi-GPU path (K sessions on one device) and device-side representations."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reset_options():
    from lightfm_amd.options import options
    defaults = dict(mode="parallel", launches_per_epoch=0, first_batch=0, max_waves=0, log_samples=False,
                    warp_kernel=0, feat_kernel=0, ramp_k=0, update_mode=0, shared_cap=0, host_positives=False)
    options.set(**defaults)
    yield
    options.set(**defaults)


def _session(model, n_items, n_users, coo, item_f=None, user_f=None, host_positives=False):
    from lightfm_amd._lightfm_fast import CSRMatrix
    from lightfm_amd.lightfm import _Session
    struct = model._get_lightfm_data()
    s = _Session(struct, CSRMatrix(item_f if item_f is not None else H.identity_features(n_items)),
                 CSRMatrix(user_f if user_f is not None else H.identity_features(n_users)))
    pos = CSRMatrix(H.positives_csr(coo)) if host_positives else None
    s.set_interactions(pos, np.ascontiguousarray(coo.row), np.ascontiguousarray(coo.col), coo.data, coo.data)
    if not host_positives:
        s.build_positives(*coo.shape)
    return s, struct


# ------------------------------------------------------------------ adadelta ---

@pytest.mark.parametrize("loss", ["warp", "bpr", "logistic"])
def test_adadelta_full_concurrency_skewed_item_stays_finite(loss):
    """adadelta's moving-average accumulators under K concurrent writers.  One item holds
    ~8 % of the interactions, the chip is full from the first launch (ramp disabled): weights must
    stay finite and the accumulators non-negative (csrc/device.hpp: publish_adadelta)."""
    from lightfm_amd import LightFM, options
    rng = np.random.RandomState(5)
    nu, ni, n = 6000, 500, 400000
    u = rng.randint(0, nu, size=n)
    i = rng.randint(0, ni, size=n)
    i[rng.rand(n) < 0.08] = 7  # the hot item
    key = np.unique(u.astype(np.int64) * ni + i)
    coo = sp.coo_matrix((np.ones(len(key), np.float32), ((key // ni).astype(np.int32), (key % ni).astype(np.int32))),
                        shape=(nu, ni), dtype=np.float32)
    options.set(ramp_k=-1)
    m = LightFM(no_components=64, loss=loss, learning_schedule="adadelta", random_state=3)
    m.fit(coo, epochs=3)
    for name in ("item_embeddings", "user_embeddings", "item_biases", "user_biases"):
        assert np.isfinite(getattr(m, name)).all(), name
    for name in ("item_embedding_gradients", "item_embedding_momentum", "item_bias_gradients",
                 "item_bias_momentum", "user_embedding_gradients", "user_embedding_momentum"):
        a = getattr(m, name)
        assert np.isfinite(a).all() and a.min() >= 0.0, name
    assert np.abs(m.item_embeddings[7]).max() < 50.0


# ------------------------------------------------------------ regularisation ---

def _labelled_problem(seed=2, nu=1500, ni=900, n_pos=60000):
    """Positive interactions from the cluster model (label +1) plus as many uniformly random pairs
    (label -1): the shape of the reference's binarised MovieLens-100k train / test sets."""
    from lightfm_amd import synthetic
    pos = synthetic.make_interactions(nu, ni, n_pos, seed=seed)
    rng = np.random.RandomState(seed + 1)
    nr, nc = rng.randint(0, nu, size=pos.nnz), rng.randint(0, ni, size=pos.nnz)
    rows = np.concatenate([pos.row, nr]).astype(np.int32)
    cols = np.concatenate([pos.col, nc]).astype(np.int32)
    vals = np.concatenate([np.ones(pos.nnz), -np.ones(pos.nnz)]).astype(np.float32)
    key, idx = np.unique(rows.astype(np.int64) * ni + cols, return_index=True)
    rows, cols, vals = rows[idx], cols[idx], vals[idx]
    mask = rng.rand(len(vals)) < 0.2

    def sub(m):
        return sp.coo_matrix((vals[m], (rows[m], cols[m])), shape=(nu, ni), dtype=np.float32)
    return sub(~mask), sub(mask)


def _auc(model, coo):
    from sklearn.metrics import roc_auc_score
    return roc_auc_score(coo.data > 0, model.predict(coo.row, coo.col))


@pytest.mark.parametrize("d", [10, 16])
@pytest.mark.parametrize("loss", ["logistic", "warp", "bpr", "warp-kos"])
def test_excessive_regularisation_parallel_mode(loss, d):
    """tests/test_movielens.py:549-569 of the reference: alpha = 1 must flatten the model (AUC near
    chance) without the lazy scale accumulating to infinity -- here in PARALLEL mode (device.hpp:
    RegScale, the boundary kernels of csrc/fit_kernels.hip).  d = 10 is the reference's default (rows padded
    to 12 floats on the device, session.hip: upload_table); both widths run the production kernels."""
    from lightfm_amd import LightFM
    train, test = _labelled_problem()
    m = LightFM(no_components=d, item_alpha=1.0, user_alpha=1.0, loss=loss, random_state=10)
    m.fit_partial(train, epochs=10, num_threads=4)
    for name in ("item_embeddings", "user_embeddings", "item_biases", "user_biases"):
        assert np.isfinite(getattr(m, name)).all(), name
    assert _auc(m, train) < 0.65
    assert _auc(m, test) < 0.65
    # a launch covers its slice of the epoch whatever alpha is (round 2 cut it after 56 interactions:
    # ~1 700 launches per epoch here); every loss runs its production kernel
    assert all(st["launches"] <= 100 for st in m._last_epoch_stats), [st["launches"] for st in m._last_epoch_stats]
    # (identity BPR / logistic: the tile kernel's regularised BPR / logistic instantiations, csrc/warp_tile_bpr.hip -- at the default
    # width too, where the narrow lane-group kernels take unregularised models only)
    want = 1 if loss in ("warp", "bpr", "logistic") else 2
    assert all(st["kernel_used"] == want for st in m._last_epoch_stats), [st["kernel_used"] for st in m._last_epoch_stats]


@pytest.mark.parametrize("loss", ["logistic", "warp"])
def test_moderate_regularisation_parallel_mode_matches_the_reference(loss):
    """tests/test_movielens.py:587-599: alpha = 1e-4, no_components=50, 30 epochs generalises (the
    unregularised model of :572-584 overfits).  The compiled reference is fit beside it on the same
    data; train / test AUC must agree within 0.02 and show the same ordering against overfitting."""
    from lightfm_amd import LightFM
    from oracle.ref_model import RefLightFM
    if not oracle.ref_available("fast"):
        pytest.skip("oracle/_ref not built")
    train, test = _labelled_problem()
    res = {}
    for name, cls in (("hip", LightFM), ("ref", RefLightFM)):
        reg = cls(no_components=50, item_alpha=0.0001, user_alpha=0.0001, random_state=10, loss=loss)
        reg.fit_partial(train, epochs=30)
        over = cls(no_components=50, random_state=10, loss=loss)
        over.fit_partial(train, epochs=30)
        res[name] = (_auc(reg, train), _auc(reg, test), _auc(over, train), _auc(over, test))
    print(res)
    for j in range(4):
        assert abs(res["hip"][j] - res["ref"][j]) < 0.02, (j, res)
    if loss == "logistic":
        assert res["hip"][2] > res["hip"][0]          # the unregularised model fits the train set better ...
        assert res["hip"][1] > res["hip"][3] - 0.005  # ... and does not generalise better


@pytest.mark.parametrize("d", [16, 10])
@pytest.mark.parametrize("layout", ["tags", "identity"])
@pytest.mark.parametrize("loss", ["logistic", "warp", "bpr"])
def test_frozen_weight_scale_folding_matches_the_oracle(loss, layout, d):
    """sample_weight = 0 freezes every gradient, so with alpha != 0 only the lazy regularisation
    acts: each visited interaction multiplies the global scale by (1 + alpha * avg_lr) and its cells
    by (1 + alpha * lr) (PYX:640-691).  Multiplications commute, so the parallel scheme (device.hpp:
    RegScale -- the live log-scale, one fold at the end) must reproduce the serial oracle -- up to what Hogwild's additive publication makes of concurrent multiplicative
    steps: k wavefronts that scale the same cell at once leave 1 + k a instead of (1 + a)^k
    (a = alpha * lr = 5e-5..1e-4 here), i.e. k a^2 / 2 per collision on the shared tag rows:
    the bar is 5e-4 relative.  d = 10: rows are 12 floats on the device; the learning-rate average of a
    step counts the 10 real components only (DModel::d_real) -- 11 / 13 of it would miss the bar by far."""
    from lightfm_amd import options
    import lightfm_amd._lightfm_fast as fast
    coo = H.make_interactions(300, 200, 8000, seed=4)
    # "tags": the row-stream kernels (feat_kernel.hpp, REG); "identity": the lane-group tile kernel (warp_tile_kernel.hpp, REG) for
    # WARP, BPR and logistic (its LOSS instantiations)
    item_f = H.tag_features(200, 12, 3, seed=1) if layout == "tags" else H.identity_features(200)
    user_f = H.identity_features(300)
    rng = np.random.RandomState(0)
    st = oracle.State(item_f.shape[1], 300, d, rng)
    a, b = st.copy(), st.copy()
    shuffle, seeds = H.epoch_inputs(coo, rng)
    zeros = np.zeros_like(coo.data)
    alpha = 0.001
    Cm = fast.CSRMatrix
    fl = fast.FastLightFM(*a.arrays(), a.d, 0, a.lr, a.rho, a.eps, a.max_sampled)
    pos = H.positives_csr(coo)
    options.set(mode="parallel", launches_per_epoch=3)
    if loss == "warp":
        fast.fit_warp(Cm(item_f), Cm(user_f), Cm(pos), coo.row, coo.col, coo.data, zeros, shuffle, fl, 0.05,
                      alpha, alpha * 2, 1, H.FixedRandom(seeds))
        oracle.fit_warp(item_f, user_f, pos, coo.row, coo.col, coo.data, zeros, shuffle, b, alpha, alpha * 2,
                        seeds, oracle.Opts(len(shuffle), rng_mode=1))
    elif loss == "bpr":
        fast.fit_bpr(Cm(item_f), Cm(user_f), Cm(pos), coo.row, coo.col, coo.data, zeros, shuffle, fl, 0.05,
                     alpha, alpha * 2, 1, H.FixedRandom(seeds))
        oracle.fit_bpr(item_f, user_f, pos, coo.row, coo.col, coo.data, zeros, shuffle, b, alpha, alpha * 2,
                       seeds, oracle.Opts(len(shuffle), rng_mode=1))
    else:
        fast.fit_logistic(Cm(item_f), Cm(user_f), coo.row, coo.col, coo.data, zeros, shuffle, fl, 0.05, alpha,
                          alpha * 2, 1)
        oracle.fit_logistic(item_f, user_f, coo.row, coo.col, coo.data, zeros, shuffle, b, alpha, alpha * 2)
    assert not np.array_equal(b.item_embeddings, st.item_embeddings)  # the regularisation did act
    # (identity: the tile kernel's REG instantiations of the three losses)
    assert options.last_kernel_used == (1 if layout == "identity" else 2)
    H.assert_states_equal(a, b, exact=False, rtol=5e-4, atol=1e-9)


# ---------------------------------------------------------- epoch segments ---

def test_epoch_as_segments_visits_every_position_once():
    """lfm_opts.pos_begin / pos_end: three segments == one epoch.  With frozen weights the per-position
    (negative, sampled) logs of the segments tile the logs of the single call exactly."""
    from lightfm_amd import LightFM
    from lightfm_amd._lightfm_fast import make_opts
    coo = H.make_interactions(400, 300, 9000, seed=6)
    m = LightFM(no_components=32, loss="warp", random_state=1)
    m._initialize(32, 300, 400)
    s, struct = _session(m, 300, 400, coo)
    try:
        # frozen: re-upload the interactions with zero sample weights
        s.set_interactions(None, np.ascontiguousarray(coo.row), np.ascontiguousarray(coo.col), coo.data,
                           np.zeros_like(coo.data))
        s.build_positives(400, 300)
        s.device_shuffle(3, 4)
        seeds = np.array([77], np.uint32)
        n = coo.nnz
        whole, wl = make_opts(n, want_log=True)
        s.epoch("warp", 0.0, 0.0, 5, 10, seeds, whole)
        neg = np.full(n, -1, np.int32)
        sampled = np.zeros(n, np.int32)
        total = np.zeros(4, np.int64)
        for b, e in ((0, 2500), (2500, 2501), (2501, n)):
            o, logs = make_opts(n, want_log=True)
            o.pos_begin, o.pos_end = b, e
            s.epoch("warp", 0.0, 0.0, 5, 10, seeds, o)
            assert o.counters[0] == e - b
            neg[b:e], sampled[b:e] = logs[0][b:e], logs[1][b:e]
            total += np.array(list(o.counters))
        assert np.array_equal(neg, wl[0]) and np.array_equal(sampled, wl[1])
        assert list(total) == list(whole.counters)
    finally:
        s.close()


# ------------------------------------------------------- device positives ---

@pytest.mark.parametrize("shape", [(50, 40, 900), (3000, 70000, 200000), (7, 5, 0)])
def test_device_positives_equal_host_tocsr(shape):
    """lfm_session_build_positives == interactions.tocsr() with sorted indices (LFM:365-372), with
    duplicate COO entries (tocsr sums them into one), empty rows and an empty matrix."""
    from lightfm_amd import LightFM
    nu, ni, n = shape
    rng = np.random.RandomState(1)
    rows = rng.randint(0, nu, size=n).astype(np.int32)
    cols = rng.randint(0, ni, size=n).astype(np.int32)
    if n:
        rows[: n // 10], cols[: n // 10] = rows[n // 10: 2 * (n // 10)], cols[n // 10: 2 * (n // 10)]  # duplicates
    coo = sp.coo_matrix((np.ones(n, np.float32), (rows, cols)), shape=(nu, ni), dtype=np.float32)
    want = coo.tocsr()
    want.sort_indices()
    m = LightFM(no_components=8, loss="warp", random_state=1)
    m._initialize(8, ni, nu)
    s, _ = _session(m, ni, nu, coo)
    try:
        indptr, indices = s.download_positives(nu)
    finally:
        s.close()
    assert np.array_equal(indptr, want.indptr)
    assert np.array_equal(indices, want.indices)


def test_fit_with_device_and_host_positives_agree():
    """Serial mode is bit-exact either way; the lookup matrices are the same matrix."""
    from lightfm_amd import LightFM, options
    coo = H.make_interactions(200, 150, 4000, seed=9)
    out = []
    for host in (False, True):
        options.set(mode="serial", host_positives=host)
        m = LightFM(no_components=16, loss="warp", random_state=4)
        m.fit(coo, epochs=2)
        out.append(m)
    options.set(host_positives=False)
    assert np.array_equal(out[0].item_embeddings, out[1].item_embeddings)
    assert np.array_equal(out[0].user_embeddings, out[1].user_embeddings)


# ------------------------------------------------------------ merge (local) ---

_MERGE_CASES = ([(m, f, "tile") for m in ("sum", "mean", "adagrad") for f in ("dense", "sparse", "overlap")]
                + [("adagrad", f, c) for f in ("sparse", "overlap") for c in ("feat", "generic")])


@pytest.mark.parametrize("mode,flavour,case", _MERGE_CASES)
def test_local_merge_matches_numpy(mode, flavour, case):
    """lfm_sessions_merge_local / _merge_local_sparse over K = 3 sessions == the numpy restatement of the
    merge the RCCL path performs (csrc/session.hip: merge_group, merge_group_sparse).  flavour: the dense
    all-reduce of whole tables; "sparse" = only the rows the epoch kernels marked dirty travel (a touched
    row that was not marked would lose its delta here); "overlap" = the sparse merge with its application
    deferred to the flush (the overlapped multi-GPU exchange).  case: which kernel family trains and marks --
    the lane-group tile kernel (WARP, identity), the row-stream kernels (BPR over [identity | tags]: shared
    rows), the generic kernels (forced; d = 10: rows of 12 floats on the device)."""
    from lightfm_amd import LightFM, _native as N
    from lightfm_amd._lightfm_fast import make_opts
    from lightfm_amd.distributed import local_shard
    from lightfm_amd.lightfm import _Session
    K, nu, ni = 3, 240, 160
    d = 10 if case == "generic" else 32
    loss = "bpr" if case == "feat" else "warp"
    item_f = H.tag_features(ni, 12, 3, seed=1) if case == "feat" else None
    n_feat = item_f.shape[1] if item_f is not None else ni
    coo = H.make_interactions(nu, ni, 9000, seed=12)
    models, sessions, structs = [], [], []
    base = LightFM(no_components=d, loss=loss, random_state=2)
    base._initialize(d, n_feat, nu)
    start = {n: getattr(base, n).copy() for n in ("item_embeddings", "item_embedding_gradients", "item_biases",
                                                  "item_bias_gradients")}
    try:
        for r in range(K):
            shard, _ = local_shard(coo, r, K)
            m = LightFM(no_components=d, loss=loss, random_state=2)
            m._initialize(d, n_feat, nu)
            s, st = _session(m, ni, nu, shard, item_f=item_f)
            s.merge_begin(1)
            s.device_shuffle(10 + r, 20 + r)
            from lightfm_amd.options import options
            options.set(warp_kernel=int(case == "generic"), feat_kernel=int(case == "generic"))
            o, _ = make_opts()
            o.history = 1 << 30
            s.epoch(loss, 0.0, 0.0, 5, 10, np.array([5 + r], np.uint32), o)
            assert int(o.kernel_used) == {"tile": 1, "feat": 2, "generic": 0}[case]
            s.sync_to_host(st)
            models.append({n: getattr(m, n).copy() for n in start})
            sessions.append(s)
            structs.append((m, st))
        if flavour == "dense":
            _Session.merge_local(sessions, 1, N.MERGE_MODES[mode])
        else:
            _Session.merge_local_sparse(sessions, 1, N.MERGE_MODES[mode], overlap=(flavour == "overlap"))
            if flavour == "overlap":  # nothing has landed yet: every replica still holds its own tables
                for (m, st), s, mm in zip(structs, sessions, models):
                    s.sync_to_host(st)
                    np.testing.assert_array_equal(m.item_embeddings, mm["item_embeddings"])
                _Session.merge_local_flush(sessions)
        for (m, st), s in zip(structs, sessions):
            s.sync_to_host(st)
    finally:
        for s in sessions:
            s.close()
    dW = [mm["item_embeddings"] - start["item_embeddings"] for mm in models]
    dG = [mm["item_embedding_gradients"] - start["item_embedding_gradients"] for mm in models]
    db = [mm["item_biases"] - start["item_biases"] for mm in models]
    dbG = [mm["item_bias_gradients"] - start["item_bias_gradients"] for mm in models]
    G = start["item_embedding_gradients"] + sum(dG)
    bG = start["item_bias_gradients"] + sum(dbG)
    if mode == "sum":
        W, b = start["item_embeddings"] + sum(dW), start["item_biases"] + sum(db)
    elif mode == "mean":
        W, b = start["item_embeddings"] + sum(dW) / K, start["item_biases"] + sum(db) / K
    else:
        def rescaled(dx, dg, g0):
            tot = sum(dg)
            return sum(x * np.sqrt((g0 + 0.5 * g) / (g0 + 0.5 * tot)) for x, g in zip(dx, dg))
        W = start["item_embeddings"] + rescaled(dW, dG, start["item_embedding_gradients"])
        b = start["item_biases"] + rescaled(db, dbG, start["item_bias_gradients"])
    assert any(np.abs(x).max() > 0 for x in dW)
    touched = np.zeros(n_feat, bool)
    for x in dG:
        touched |= np.abs(x).max(axis=1) > 0
    assert 0 < touched.sum()
    for m, _ in structs:  # every replica holds the merged item tables; user tables are untouched
        np.testing.assert_allclose(m.item_embeddings, W, rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(m.item_embedding_gradients, G, rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(m.item_biases, b, rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(m.item_bias_gradients, bG, rtol=2e-5, atol=1e-7)
        # rows nobody touched are bit-for-bit what they were
        np.testing.assert_array_equal(m.item_embeddings[~touched], start["item_embeddings"][~touched])
    if flavour != "overlap":  # a merge that has landed leaves all replicas bit-identical
        for m, _ in structs[1:]:
            np.testing.assert_array_equal(m.item_embeddings, structs[0][0].item_embeddings)
            np.testing.assert_array_equal(m.item_embedding_gradients, structs[0][0].item_embedding_gradients)


def test_hot_rows_merge_between_full_merges():
    """lfm_session_set_hot_rows / lfm_sessions_merge_local_hot: the shared tag rows of a hybrid model are merged
    on their own (every replica then agrees on them, and holds its own values everywhere else); the next full
    merge exchanges the rest and leaves the hot rows where they are.  SUM mode against numpy."""
    from lightfm_amd import LightFM, _native as N
    from lightfm_amd._lightfm_fast import make_opts
    from lightfm_amd.distributed import hot_rows, local_shard
    from lightfm_amd.lightfm import _Session
    K, nu, ni, d = 3, 240, 160, 32
    item_f = H.tag_features(ni, 12, 3, seed=1)
    n_feat = item_f.shape[1]
    hot = hot_rows(item_f, 1.0 / 512)
    assert 0 < len(hot) <= 12 and hot.min() >= ni  # the tag columns, not the identity block
    cold = np.setdiff1d(np.arange(n_feat), hot)
    coo = H.make_interactions(nu, ni, 9000, seed=12)
    base = LightFM(no_components=d, loss="bpr", random_state=2)
    base._initialize(d, n_feat, nu)
    W0, G0 = base.item_embeddings.copy(), base.item_embedding_gradients.copy()
    own, sessions, structs = [], [], []
    try:
        for r in range(K):
            shard, _ = local_shard(coo, r, K)
            m = LightFM(no_components=d, loss="bpr", random_state=2)
            m._initialize(d, n_feat, nu)
            s, st = _session(m, ni, nu, shard, item_f=item_f)
            s.merge_begin(1)
            s.set_hot_rows(0, hot)
            s.device_shuffle(10 + r, 20 + r)
            o, _ = make_opts()
            o.history = 1 << 30
            s.epoch("bpr", 0.0, 0.0, 5, 10, np.array([5 + r], np.uint32), o)
            s.sync_to_host(st)
            own.append((m.item_embeddings.copy(), m.item_embedding_gradients.copy()))
            sessions.append(s)
            structs.append((m, st))
        W = W0 + sum(w - W0 for w, _ in own)
        G = G0 + sum(g - G0 for _, g in own)
        _Session.merge_local_hot(sessions, 1, N.MERGE_MODES["sum"])
        for ((m, st), s), (w, g) in zip(zip(structs, sessions), own):
            s.sync_to_host(st)
            np.testing.assert_allclose(m.item_embeddings[hot], W[hot], rtol=2e-5, atol=1e-7)
            np.testing.assert_allclose(m.item_embedding_gradients[hot], G[hot], rtol=2e-5, atol=1e-7)
            np.testing.assert_array_equal(m.item_embeddings[cold], w[cold])  # untouched by the hot merge
        hot_after = structs[0][0].item_embeddings[hot].copy()
        _Session.merge_local_sparse(sessions, 1, N.MERGE_MODES["sum"])
        for (m, st), s in zip(structs, sessions):
            s.sync_to_host(st)
            np.testing.assert_allclose(m.item_embeddings, W, rtol=2e-5, atol=1e-7)
            np.testing.assert_allclose(m.item_embedding_gradients, G, rtol=2e-5, atol=1e-7)
            np.testing.assert_array_equal(m.item_embeddings[hot], hot_after)  # already exchanged: nothing to add
        with pytest.raises(ValueError):
            sessions[0].set_hot_rows(0, np.array([3, 2], np.int32))  # not ascending
        with pytest.raises(ValueError):
            sessions[0].set_hot_rows(0, np.array([n_feat], np.int32))  # not a feature row
    finally:
        for s in sessions:
            s.close()


_ITEM_TABLES = ("item_embeddings", "item_embedding_gradients", "item_embedding_momentum", "item_biases",
                "item_bias_gradients", "item_bias_momentum")


@pytest.mark.parametrize("schedule,mode", [("adagrad", "sum"), ("adagrad", "mean"), ("adagrad", "adagrad"),
                                           ("adadelta", "sum"), ("adadelta", "mean"), ("adadelta", "adagrad")])
def test_merge_flavours_agree_from_identical_states(schedule, mode):
    """The dense merge, the sparse merge (detected rows) and the sparse merge over ALL rows (what follows a union of
    >= 90 % of a table: no detection, no OR, no compaction -- lfm_session_set_merge_dense_fraction) applied to the SAME
    snapshot and the SAME trained replicas (lfm_session_load_model puts them back): the all-rows merge is bit-identical
    to the detected-rows merge, both equal the dense merge.  adadelta: the sparse merge carries the momentum tables too."""
    from lightfm_amd import LightFM, _native as N
    from lightfm_amd._lightfm_fast import make_opts
    from lightfm_amd.distributed import local_shard
    from lightfm_amd.lightfm import _Session
    K, nu, ni, d = 3, 240, 20000, 32
    coo = H.make_interactions(nu, ni, 5000, seed=12, zipf=1.1)   # (a long catalogue: many rows stay untouched)
    sessions, structs, trained = [], [], []
    names = [n for n in _ITEM_TABLES if schedule == "adadelta" or "momentum" not in n]
    try:
        for r in range(K):
            shard, _ = local_shard(coo, r, K)
            m = LightFM(no_components=d, loss="warp", random_state=2, learning_schedule=schedule)
            m._initialize(d, ni, nu)
            s, st = _session(m, ni, nu, shard)
            start = {n: getattr(m, n).copy() for n in _ITEM_TABLES}
            s.device_shuffle(10 + r, 20 + r)
            o, _ = make_opts()
            o.history = 1 << 30
            s.epoch("warp", 0.0, 0.0, 5, 10, np.array([5 + r], np.uint32), o)
            s.sync_to_host(st)
            trained.append({n: getattr(m, n).copy() for n in _ITEM_TABLES})
            sessions.append(s)
            structs.append((m, st))
        results = {}
        for flavour in ("dense", "sparse", "allrows"):
            for r, ((m, st), s) in enumerate(zip(structs, sessions)):
                for n in _ITEM_TABLES:
                    getattr(m, n)[...] = start[n]
                s.load_model(st)
                s.set_merge_dense_fraction(0.0 if flavour == "allrows" else 2.0)
                s.merge_begin(1)            # the interval starts at the common initial state ...
                for n in _ITEM_TABLES:
                    getattr(m, n)[...] = trained[r][n]
                s.load_model(st)            # ... and every replica holds what it trained
            if flavour == "dense":
                _Session.merge_local(sessions, 1, N.MERGE_MODES[mode])
            else:
                _Session.merge_local_sparse(sessions, 1, N.MERGE_MODES[mode])
            out = []
            for (m, st), s in zip(structs, sessions):
                s.sync_to_host(st)
                out.append({n: getattr(m, n).copy() for n in names})
            for other in out[1:]:
                for n in names:
                    np.testing.assert_array_equal(other[n], out[0][n])
            results[flavour] = out[0]
    finally:
        for s in sessions:
            s.close()
    touched = np.zeros(ni, bool)
    for t in trained:
        touched |= np.any(t["item_embedding_gradients"] != start["item_embedding_gradients"], axis=1)
    assert 0.05 < touched.mean() < 0.9, touched.mean()
    for n in names:
        np.testing.assert_array_equal(results["allrows"][n], results["sparse"][n], err_msg=n)
        np.testing.assert_allclose(results["sparse"][n], results["dense"][n], rtol=2e-6, atol=1e-7, err_msg=n)
        assert np.array_equal(results["sparse"][n][~touched], start[n][~touched]), n
    assert not np.array_equal(results["sparse"]["item_embeddings"], start["item_embeddings"])


def test_sparse_merge_carries_what_training_adds_while_the_exchange_is_in_flight():
    """The overlapped exchange: a second segment trains between the merge call and the flush.  What it adds to
    the local tables must survive the late application (table += sum - local delta) and travel with the NEXT
    merge: after two merges and the final flush every replica holds start + all deltas of both segments."""
    from lightfm_amd import LightFM, _native as N
    from lightfm_amd._lightfm_fast import make_opts
    from lightfm_amd.distributed import local_shard
    from lightfm_amd.lightfm import _Session
    K, nu, ni, d = 2, 200, 120, 16
    coo = H.make_interactions(nu, ni, 6000, seed=7)
    sessions, structs = [], []
    base = LightFM(no_components=d, loss="warp", random_state=2)
    base._initialize(d, ni, nu)
    G0 = base.item_embedding_gradients.copy()
    per_segment = []
    try:
        for r in range(K):
            shard, _ = local_shard(coo, r, K)
            m = LightFM(no_components=d, loss="warp", random_state=2)
            m._initialize(d, ni, nu)
            s, st = _session(m, ni, nu, shard)
            s.merge_begin(1)
            s.device_shuffle(3 + r, 4 + r)
            sessions.append(s)
            structs.append((m, st, shard.nnz))
        seen = [G0.copy() for _ in range(K)]
        for seg in range(2):
            dG = []
            for r, ((m, st, n), s) in enumerate(zip(structs, sessions)):
                o, _ = make_opts()
                o.history = 1 << 30
                o.pos_begin, o.pos_end = (0, n // 2) if seg == 0 else (n // 2, n)
                s.epoch("warp", 0.0, 0.0, 5, 10, np.array([9 + r], np.uint32), o)
                s.sync_to_host(st)
                # what THIS segment added locally (accumulators only grow; SUM mode keeps them additive)
                dG.append(m.item_embedding_gradients - seen[r])
            per_segment.append(dG)
            _Session.merge_local_sparse(sessions, 1, N.MERGE_MODES["sum"], overlap=True)
            for r, ((m, st, n), s) in enumerate(zip(structs, sessions)):
                s.sync_to_host(st)
                seen[r] = m.item_embedding_gradients.copy()  # segment 0's exchange lands inside the next merge call
        _Session.merge_local_flush(sessions)
        for (m, st, n), s in zip(structs, sessions):
            s.sync_to_host(st)
    finally:
        for s in sessions:
            s.close()
    # accumulators: start + every rank's growth of both segments.  `seen` bookkeeping: after merge call 1 nothing
    # has landed (seen = own table); after merge call 2 segment 0's sum has landed.
    total = G0 + sum(per_segment[0]) + per_segment[1][0] + per_segment[1][1]
    # per_segment[1][r] was measured against a table that did not yet hold the other rank's segment-0 growth
    for m, _, _ in structs:
        np.testing.assert_allclose(m.item_embedding_gradients, total, rtol=2e-5, atol=1e-6)


def test_load_model_roundtrip():
    from lightfm_amd import LightFM
    coo = H.make_interactions(100, 80, 1500, seed=3)
    m = LightFM(no_components=16, loss="warp", random_state=1)
    m._initialize(16, 80, 100)
    s, st = _session(m, 80, 100, coo)
    try:
        want = m.item_embeddings.copy() * 3.0
        m.item_embeddings[...] = want
        s.load_model(st)
        m.item_embeddings[...] = 0
        s.sync_to_host(st)
    finally:
        s.close()
    assert np.array_equal(m.item_embeddings, want)


# ---------------------------------------------------------- representations ---

@pytest.mark.parametrize("d", [10, 64, 130])
def test_representations_match_scipy_products(d):
    """get_item_representations / get_user_representations with a feature matrix (LFM:991-1047) ==
    the reference's `features * biases`, `features * embeddings`."""
    from lightfm_amd import LightFM
    coo = H.make_interactions(120, 90, 2500, seed=2)
    item_f = H.tag_features(90, 25, 4, seed=3)
    user_f = H.tag_features(120, 10, 2, seed=4, normalise=True)
    m = LightFM(no_components=d, loss="warp", random_state=5)
    m.fit(coo, item_features=item_f, user_features=user_f, epochs=1)
    for feats, getter, emb, bias in ((item_f, m.get_item_representations, m.item_embeddings, m.item_biases),
                                     (user_f, m.get_user_representations, m.user_embeddings, m.user_biases)):
        b, e = getter(feats)
        assert e.shape == (feats.shape[0], d) and b.shape == (feats.shape[0],)
        assert e.dtype == np.float32 and b.dtype == np.float32
        np.testing.assert_allclose(e, feats * emb, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(b, feats * bias, rtol=1e-5, atol=1e-7)
        b0, e0 = getter()
        assert b0 is bias and e0 is emb
    with pytest.raises(ValueError):
        m.get_item_representations(sp.identity(7, dtype=np.float32, format="csr"))
