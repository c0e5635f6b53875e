"""GPU parity tests of the pipelined row-stream kernels (lightfm_amd/csrc/feat_kernel.hpp): parallel
mode for feature CSRs, BPR, k-OS WARP and logistic (BASELINE configs C3 / C5), against the CPU
oracle with one PRNG stream per shuffled position (the rule both sides share).

Bars:
  * frozen weights (sample_weight = 0, the reference's own trick, tests/test_movielens.py:517-533):
    chosen negative and sample count of EVERY position exact, totals of draws / updates /
    in_positives probes exact, weights untouched -- WARP and BPR, identity and tag features on
    either side, d = 8 .. 128, single- and multi-batch max_sampled;
  * one interaction per launch (launches_per_epoch = n): the Hogwild kernel is then sequential and
    logs, weights, biases and accumulators must equal the oracle's -- all four losses (k-OS
    included, which cannot be weight-frozen);
  * the generic kernels (feat_kernel = 1) pass the same checks;
  * full Hogwild training fits like the sequential oracle.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu

_DEFAULTS = dict(mode="parallel", launches_per_epoch=0, first_batch=0, max_waves=0, log_samples=False,
                 warp_kernel=0, feat_kernel=0, update_mode=0, debug=0, ramp_k=0, shared_cap=0)


@pytest.fixture(scope="module")
def fast():
    import lightfm_amd._lightfm_fast as f
    from lightfm_amd import _native
    assert _native.device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    return f


@pytest.fixture(autouse=True)
def _reset_options():
    from lightfm_amd.options import options
    options.set(**_DEFAULTS)
    yield
    options.set(**_DEFAULTS)


def _features(kind, n, seed):
    if kind == "id":
        return H.identity_features(n)
    if kind == "tags":      # [identity | 3 weighted tags of 12]
        return H.tag_features(n, 12, 3, seed)
    if kind == "tagsonly":  # no identity block: 4 of 30 shared columns, L1-normalised rows
        return H.tag_features(n, 30, 4, seed, with_identity=False, normalise=True)
    if kind == "wide":      # rows of ~70 entries: longer than one wavefront of lanes
        return H.tag_features(n, 400, 70, seed)
    raise ValueError(kind)


def _spread(st, f_item, f_user):
    """Scale the fresh embeddings so that scores have a standard deviation of a few units: the
    margin test (PYX:875) then sees both outcomes."""
    a = 3.0 / st.d ** 0.25
    st.item_embeddings *= 2 * st.d * a / np.sqrt(max(1.0, f_item))
    st.user_embeddings *= 2 * st.d * a / np.sqrt(max(1.0, f_user))


def _run_hip(fast, loss, coo, item_f, user_f, st, shuffle, seeds, weight, k=3, n=5):
    Cm = fast.CSRMatrix
    fl = fast.FastLightFM(*st.arrays(), st.d, int(st.schedule == "adadelta"), st.lr, st.rho, st.eps,
                          st.max_sampled)
    pos = H.positives_csr(coo)
    rs = H.FixedRandom(seeds)
    if loss == "warp":
        fast.fit_warp(Cm(item_f), Cm(user_f), Cm(pos), coo.row, coo.col, coo.data, weight, shuffle, fl, 0.05,
                      0.0, 0.0, len(seeds), rs)
    elif loss == "bpr":
        fast.fit_bpr(Cm(item_f), Cm(user_f), Cm(pos), coo.row, coo.col, coo.data, weight, shuffle, fl, 0.05,
                     0.0, 0.0, len(seeds), rs)
    elif loss == "warp-kos":
        fast.fit_warp_kos(Cm(item_f), Cm(user_f), Cm(pos), coo.row, shuffle, fl, 0.05, 0.0, 0.0, k, n,
                          len(seeds), rs)
    else:
        fast.fit_logistic(Cm(item_f), Cm(user_f), coo.row, coo.col, coo.data, weight, shuffle, fl, 0.05, 0.0,
                          0.0, 1)


def _run_orc(loss, coo, item_f, user_f, st, shuffle, seeds, weight, k=3, n=5):
    o = oracle.Opts(len(shuffle), rng_mode=1, log=True)
    pos = H.positives_csr(coo)
    if loss == "warp":
        oracle.fit_warp(item_f, user_f, pos, coo.row, coo.col, coo.data, weight, shuffle, st, 0.0, 0.0, seeds, o)
    elif loss == "bpr":
        oracle.fit_bpr(item_f, user_f, pos, coo.row, coo.col, coo.data, weight, shuffle, st, 0.0, 0.0, seeds, o)
    elif loss == "warp-kos":
        oracle.fit_warp_kos(item_f, user_f, pos, coo.row, shuffle, st, 0.0, 0.0, k, n, seeds, o)
    else:
        oracle.fit_logistic(item_f, user_f, coo.row, coo.col, coo.data, weight, shuffle, st, 0.0, 0.0, o)
    return o


FROZEN = [
    # (id, n_users, n_items, nnz, d, max_sampled, first_batch, item feats, user feats, ratings)
    ("d64-tags", 300, 200, 6000, 64, 10, 0, "tags", "id", False),
    ("d128-tags-fb3", 200, 150, 4000, 128, 10, 3, "tags", "id", True),
    ("d128-tagsonly-both", 120, 90, 2500, 128, 10, 0, "tagsonly", "tags", False),
    ("d8-tags-ms7", 60, 40, 500, 8, 7, 0, "tags", "tags", True),
    ("d96-tags-ms40-multibatch", 100, 300, 3000, 96, 40, 0, "tags", "id", False),
    ("d32-wide-rows", 80, 60, 1500, 32, 10, 0, "wide", "id", False),
    ("d64-identity", 150, 100, 3000, 64, 10, 0, "id", "id", True),
    ("d20-user-tags-only", 90, 70, 1200, 20, 12, 5, "id", "tagsonly", False),
    # widths that are no multiple of 4 (rows padded on the device) and 128 < d <= 256 (four components per lane)
    ("d10-tags", 150, 100, 3000, 10, 10, 0, "tags", "id", True),
    ("d30-tagsonly-both", 120, 90, 2500, 30, 10, 0, "tagsonly", "tags", False),
    ("d50-tags", 100, 80, 2000, 50, 10, 0, "tags", "id", False),
    ("d200-tags", 100, 80, 2000, 200, 10, 0, "tags", "id", True),
    ("d256-tagsonly-both", 60, 50, 900, 256, 10, 0, "tagsonly", "tags", False),
]


@pytest.mark.parametrize("case", FROZEN, ids=[c[0] for c in FROZEN])
@pytest.mark.parametrize("loss", ["warp", "bpr"])
@pytest.mark.parametrize("kernel", ["feat", "generic"])
def test_frozen_weights_samples_exact(fast, case, loss, kernel):
    from lightfm_amd.options import options
    _, nu, ni, nnz, d, ms, fb, itf, usf, ratings = case
    coo = H.make_interactions(nu, ni, nnz, seed=17, ratings=ratings, zipf=0.6)
    item_f, user_f = _features(itf, ni, 11), _features(usf, nu, 13)
    rng = np.random.RandomState(9)
    st = oracle.State(item_f.shape[1], user_f.shape[1], d, rng, max_sampled=ms)
    _spread(st, item_f.nnz / ni, user_f.nnz / nu)
    st.item_biases[:] = rng.randn(item_f.shape[1]).astype(np.float32) * 0.3
    st.user_biases[:] = rng.randn(user_f.shape[1]).astype(np.float32) * 0.3
    a, b = st.copy(), st.copy()
    zeros = np.zeros_like(coo.data)
    shuffle, seeds = H.epoch_inputs(coo, rng)
    # warp_kernel = 1 keeps identity-feature WARP away from the lane-group tile kernel
    options.set(log_samples=True, launches_per_epoch=3, first_batch=fb, warp_kernel=1,
                feat_kernel=1 if kernel == "generic" else 0)
    _run_hip(fast, loss, coo, item_f, user_f, a, shuffle, seeds, zeros)
    o = _run_orc(loss, coo, item_f, user_f, b, shuffle, seeds, zeros)
    neg, sampled = options.last_logs
    assert np.array_equal(sampled, o.sampled), "sample counts differ"
    assert np.array_equal(neg, o.neg), "negative (rank) indices differ"
    assert options.last_counters == o.counters
    assert o.counters[2] > 0, "no update: the case does not exercise the update path"
    H.assert_states_equal(a, st, exact=True)
    assert options.last_kernel_used == (0 if kernel == "generic" else 2), (kernel, options.last_kernel_used)


SEQ = [
    # (id, d, max_sampled, item feats, user feats)
    ("d64-tags", 64, 10, "tags", "id"),
    ("d128-tagsonly-both", 128, 10, "tagsonly", "tags"),
    ("d20-tags-ms20", 20, 20, "tags", "tags"),
    ("d32-wide", 32, 6, "wide", "id"),
    ("d128-identity", 128, 10, "id", "id"),
    ("d10-tags", 10, 10, "tags", "id"),
    ("d50-tags-both", 50, 10, "tags", "tags"),
    ("d200-tags", 200, 10, "tags", "id"),
]


@pytest.mark.parametrize("case", SEQ, ids=[c[0] for c in SEQ])
@pytest.mark.parametrize("loss", ["warp", "bpr", "logistic", "warp-kos"])
@pytest.mark.parametrize("update_mode", [1, 3], ids=["store", "atomic"])
def test_one_interaction_per_launch_matches_the_oracle(fast, case, loss, update_mode):
    """launches_per_epoch = n makes the Hogwild kernel sequential: the sample logs equal the oracle's
    (same per-position streams) and, two epochs later, so do all twelve arrays -- bit for bit with
    plain stores for WARP / k-OS (BPR / logistic call exp() of the device libm: 1e-6); with atomic
    deltas the adder of the atomic unit replaces the reference's rounding: 2e-4."""
    from lightfm_amd.options import options
    _, d, ms, itf, usf = case
    nu, ni = 40, 30
    coo = H.make_interactions(nu, ni, 260, seed=3, ratings=(loss != "warp-kos"))
    item_f, user_f = _features(itf, ni, 5), _features(usf, nu, 6)
    rng = np.random.RandomState(4)
    st = oracle.State(item_f.shape[1], user_f.shape[1], d, rng, max_sampled=ms)
    _spread(st, item_f.nnz / ni, user_f.nnz / nu)
    a, b = st.copy(), st.copy()
    options.set(log_samples=(loss != "logistic"), launches_per_epoch=len(coo.data), update_mode=update_mode,
                warp_kernel=1)
    weight = coo.data if loss != "logistic" else np.ones_like(coo.data)
    for _ in range(2):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _run_hip(fast, loss, coo, item_f, user_f, a, shuffle, seeds, weight)
        o = _run_orc(loss, coo, item_f, user_f, b, shuffle, seeds, weight)
        if loss != "logistic":
            neg, sampled = options.last_logs
            assert np.array_equal(sampled, o.sampled)
            assert np.array_equal(neg, o.neg)
        assert options.last_counters == o.counters
    assert not np.array_equal(a.item_embeddings, st.item_embeddings)
    assert options.last_kernel_used == 2, "the row-stream kernels did not run"
    if update_mode == 1 and loss in ("warp", "warp-kos"):
        H.assert_states_equal(a, b, exact=True)
    elif update_mode == 1:
        H.assert_states_equal(a, b, exact=False, rtol=2e-6, atol=1e-6)
    else:
        # the float adder of the L2 atomic unit does not round like v_add_f32: a cell updated ~100
        # times (the shared tag rows) drifts by tens of ulps
        H.assert_states_equal(a, b, exact=False, rtol=2e-4, atol=5e-6)


@pytest.mark.parametrize("n,k,d,itf", [(40, 7, 16, "id"), (63, 63, 32, "tags"), (64, 5, 16, "id")],
                         ids=["n40-k7-d16", "n63-k63-d32-tags", "n64-generic"])
def test_kos_many_sampled_positives(fast, n, k, d, itf):
    """fit_warp_kos with n up to 63 on the row-stream kernel (one job per lane: the user + n sampled positives; round 5 stopped
    at 32); n = 64 runs the generic kernel.  One interaction per launch, plain stores: bit for bit the oracle."""
    from lightfm_amd.options import options
    nu, ni = 12, 150
    rng = np.random.RandomState(n)
    dense = rng.rand(nu, ni) < 0.6
    dense[3, :] = False
    dense[3, :5] = True  # (a user with fewer positives than n: no_pos = 5, PYX:975)
    m = sp.coo_matrix(dense.astype(np.float32))
    coo = sp.coo_matrix((np.ones(m.nnz, np.float32), (m.row.astype(np.int32), m.col.astype(np.int32))), shape=(nu, ni), dtype=np.float32)
    item_f, user_f = _features(itf, ni, 5), _features("id", nu, 6)
    st = oracle.State(item_f.shape[1], user_f.shape[1], d, rng, max_sampled=10)
    _spread(st, item_f.nnz / ni, 1.0)
    a, b = st.copy(), st.copy()
    options.set(log_samples=True, launches_per_epoch=coo.nnz, update_mode=1, warp_kernel=1)
    for _ in range(2):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _run_hip(fast, "warp-kos", coo, item_f, user_f, a, shuffle, seeds, coo.data, k=k, n=n)
        o = _run_orc("warp-kos", coo, item_f, user_f, b, shuffle, seeds, coo.data, k=k, n=n)
        neg, sampled = options.last_logs
        assert np.array_equal(sampled, o.sampled) and np.array_equal(neg, o.neg)
        assert options.last_counters == o.counters
        assert options.last_kernel_used == (2 if n <= 63 else 0), options.last_kernel_used
    assert not np.array_equal(a.item_embeddings, st.item_embeddings)
    H.assert_states_equal(a, b, exact=True)


@pytest.mark.parametrize("loss", ["bpr", "warp-kos", "logistic", "warp"])
def test_feat_training_learns_like_the_oracle(fast, loss):
    """Full Hogwild training through the row-stream kernels (tag features, d = 128): fit quality
    close to the sequential oracle's after 4 epochs (neither side is order-deterministic)."""
    nu, ni = 1500, 1000
    coo = H.make_interactions(nu, ni, 60000, seed=21, ratings=(loss == "logistic"))
    item_f, user_f = _features("tags", ni, 3), H.identity_features(nu)
    rng = np.random.RandomState(5)
    st = oracle.State(item_f.shape[1], nu, 128, rng)
    a, b = st.copy(), st.copy()
    weight = coo.data if loss != "logistic" else np.ones_like(coo.data)
    for _ in range(4):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _run_hip(fast, loss, coo, item_f, user_f, a, shuffle, seeds, weight, k=5, n=10)
        _run_orc(loss, coo, item_f, user_f, b, shuffle, seeds, weight, k=5, n=10)
    from lightfm_amd.options import options
    assert options.last_counters[2] > 0

    keep = coo.data > 0

    def margin(s):
        r = np.random.RandomState(0)
        pos = oracle.predict(item_f, user_f, coo.row[keep], coo.col[keep], s)
        neg = oracle.predict(item_f, user_f, coo.row[keep],
                             r.randint(0, ni, size=int(keep.sum())).astype(np.int32), s)
        return float(np.mean(pos - neg)), float(np.mean(pos > neg))

    (ma, aa), (mb, ab) = margin(a), margin(b)
    assert ab > (0.58 if loss == "logistic" else 0.7)  # logistic sees the ratings as noisy labels
    assert abs(aa - ab) < 0.04, (aa, ab)  # Hogwild: run-to-run variation
    assert abs(ma - mb) / abs(mb) < 0.25, (ma, mb)


def test_feat_kernel_is_the_one_that_ran(fast):
    """lfm_opts.kernel_used: 2 = row-stream kernels, 0 = generic (forced), 1 = lane-group tile kernel."""
    from lightfm_amd import LightFM, options
    coo = H.make_interactions(300, 200, 5000, seed=2)
    feats = H.tag_features(200, 12, 3, 1)
    seen = {}
    for name, kw, model_kw, f in (("feat", {}, dict(loss="bpr"), feats),
                                  ("generic", dict(feat_kernel=1), dict(loss="bpr"), feats),
                                  ("tile", {}, dict(loss="warp"), None),
                                  ("adadelta", {}, dict(loss="bpr", learning_schedule="adadelta"), feats)):
        options.set(**_DEFAULTS)
        options.set(**kw)
        m = LightFM(no_components=64, random_state=1, **model_kw)
        m.fit(coo, item_features=f, epochs=1)
        seen[name] = m._last_epoch_stats[-1]["kernel_used"]
    assert seen == {"feat": 2, "generic": 0, "tile": 1, "adadelta": 2}  # (adadelta: the ADA instantiations since round 6)


ADA_SEQ = [
    ("d64-tags", 64, 10, "tags", "id"),
    ("d128-tagsonly-both", 128, 10, "tagsonly", "tags"),
    ("d10-tags", 10, 10, "tags", "id"),
    ("d32-wide", 32, 6, "wide", "id"),
]


@pytest.mark.parametrize("case", ADA_SEQ, ids=[c[0] for c in ADA_SEQ])
@pytest.mark.parametrize("loss", ["warp", "bpr", "logistic", "warp-kos"])
@pytest.mark.parametrize("update_mode", [1, 3], ids=["store", "atomic"])
def test_adadelta_one_interaction_per_launch_matches_the_oracle(fast, case, loss, update_mode):
    """The adadelta schedule (PYX:416-434) on the row-stream kernels (ADA instantiations: the momentum rows travel with W
    and G, the moving averages are published by compare-and-swap): sequential launches against the oracle, the bars of
    the adagrad test above."""
    from lightfm_amd.options import options
    _, d, ms, itf, usf = case
    nu, ni = 40, 30
    coo = H.make_interactions(nu, ni, 260, seed=3, ratings=(loss != "warp-kos"))
    item_f, user_f = _features(itf, ni, 5), _features(usf, nu, 6)
    rng = np.random.RandomState(4)
    st = oracle.State(item_f.shape[1], user_f.shape[1], d, rng, schedule="adadelta", max_sampled=ms)
    _spread(st, item_f.nnz / ni, user_f.nnz / nu)
    a, b = st.copy(), st.copy()
    options.set(log_samples=(loss != "logistic"), launches_per_epoch=len(coo.data), update_mode=update_mode, warp_kernel=1)
    weight = coo.data if loss != "logistic" else np.ones_like(coo.data)
    for _ in range(2):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _run_hip(fast, loss, coo, item_f, user_f, a, shuffle, seeds, weight)
        o = _run_orc(loss, coo, item_f, user_f, b, shuffle, seeds, weight)
        if loss != "logistic":
            neg, sampled = options.last_logs
            assert np.array_equal(sampled, o.sampled) and np.array_equal(neg, o.neg)
        assert options.last_counters == o.counters
    assert options.last_kernel_used == 2, "the row-stream kernels did not run"
    assert not np.array_equal(a.item_embedding_momentum, st.item_embedding_momentum), "the momentum tables were not trained"
    if update_mode == 1 and loss in ("warp", "warp-kos"):
        H.assert_states_equal(a, b, exact=True)
    elif update_mode == 1:
        H.assert_states_equal(a, b, exact=False, rtol=2e-6, atol=1e-6)
    else:
        H.assert_states_equal(a, b, exact=False, rtol=2e-4, atol=5e-6)


@pytest.mark.parametrize("loss", ["bpr", "warp-kos", "warp"])
def test_adadelta_hogwild_on_the_row_stream_kernels_learns(loss):
    """Full concurrency, hybrid model, adadelta: finite tables (the compare-and-swap publication keeps the moving
    averages positive under any number of concurrent writers of a shared row) and a model that ranks its positives."""
    from lightfm_amd import LightFM, options
    coo = H.make_interactions(3000, 2000, 150_000, seed=8, zipf=0.8)
    feats = H.tag_features(2000, 30, 3, 2)
    m = LightFM(no_components=64, loss=loss, learning_schedule="adadelta", random_state=3)
    m.fit(coo, item_features=feats, epochs=4)
    assert m._last_epoch_stats[-1]["kernel_used"] == 2
    for name in ("item_embeddings", "item_embedding_gradients", "item_embedding_momentum", "user_embeddings", "item_biases"):
        assert np.isfinite(getattr(m, name)).all(), name
    assert (m.item_embedding_gradients >= 0).all() and (m.item_embedding_momentum >= 0).all()
    rows, cols = np.ascontiguousarray(coo.row), np.ascontiguousarray(coo.col)
    negs = np.random.RandomState(0).randint(0, 2000, size=coo.nnz).astype(np.int32)
    acc = float(np.mean(m.predict(rows, cols, item_features=feats) > m.predict(rows, negs, item_features=feats)))
    assert acc > 0.75, acc
