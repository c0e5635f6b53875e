"""GPU parity tests of the hot set (lightfm_amd/csrc/hot_slices.hip; the HOT instantiations of csrc/feat_kernel.hpp):
shared item-feature rows -- the tag rows of a hybrid model, BASELINE config C3 -- accumulated in LDS component slices
between launches instead of by float atomics of every interaction.  Against the CPU oracle (PYX:454-649):

  * one interaction per launch: the two-phase path is then sequential (one replica, one record) and weights, biases and
    accumulators must equal the oracle's within the bar of every atomically published update (old + fl32(new - old): a few
    float32 ulps of the array's largest magnitude, <= 8 here), samples and counters exactly -- all four losses, weighted tags, a tag
    shared by the positive and the negative item included, d = 16 / 40 / 128;
  * frozen weights under full concurrency and the default launch plan: every position's negative and draw count and the
    counters equal the oracle's, no table moves (a record with gradient 0 leaves its LDS cells as they were);
  * full-concurrency training: the model learns as it does with the rows on the float atomics (lfm_opts.debug bit 14).
`lfm_opts.plan_flags` bit 5 asserts that the hot set was in use.
"""
import numpy as np
import pytest

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


def _run(fast, loss, coo, item_f, user_f, st, shuffle, seeds, weight, k=3, n=5):
    Cm = fast.CSRMatrix
    fl = fast.FastLightFM(*st.arrays(), st.d, 0, st.lr, st.rho, st.eps, st.max_sampled)
    pos = H.positives_csr(coo)
    rs = H.FixedRandom(seeds)
    if loss == "warp":
        fast.fit_warp(Cm(item_f), Cm(user_f), Cm(pos), coo.row, coo.col, coo.data, weight, shuffle, fl, 0.05, 0.0, 0.0, len(seeds), rs)
    elif loss == "bpr":
        fast.fit_bpr(Cm(item_f), Cm(user_f), Cm(pos), coo.row, coo.col, coo.data, weight, shuffle, fl, 0.05, 0.0, 0.0, len(seeds), rs)
    elif loss == "warp-kos":
        fast.fit_warp_kos(Cm(item_f), Cm(user_f), Cm(pos), coo.row, shuffle, fl, 0.05, 0.0, 0.0, k, n, len(seeds), rs)
    else:
        fast.fit_logistic(Cm(item_f), Cm(user_f), coo.row, coo.col, coo.data, weight, shuffle, fl, 0.05, 0.0, 0.0, 1)


def _orc(loss, coo, item_f, user_f, st, shuffle, seeds, weight, k=3, n=5):
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


def _state(item_f, user_f, d, rng, ms=10):
    st = oracle.State(item_f.shape[1], user_f.shape[1], d, rng, max_sampled=ms)
    a = 3.0 / d ** 0.25
    st.item_embeddings *= 2 * d * a / np.sqrt(item_f.nnz / item_f.shape[0])
    st.user_embeddings *= 2 * d * a / np.sqrt(user_f.nnz / user_f.shape[0])
    st.item_biases[:] = rng.randn(item_f.shape[1]).astype(np.float32) * 0.3
    st.user_biases[:] = rng.randn(user_f.shape[1]).astype(np.float32) * 0.3
    return st


SEQ = [
    # (id, n_users, n_items, nnz, d, n_tags, tags per item, user features)
    ("d16-3of6", 60, 50, 500, 16, 6, 3, "id"),
    ("d128-8of40", 50, 60, 400, 128, 40, 8, "id"),
    ("d40-4of9-user-tags", 50, 40, 400, 40, 9, 4, "tags"),
]


@pytest.mark.parametrize("case", SEQ, ids=[c[0] for c in SEQ])
@pytest.mark.parametrize("loss", ["warp", "bpr", "logistic", "warp-kos"])
def test_one_interaction_per_launch_matches_the_oracle(fast, case, loss):
    from lightfm_amd.options import options
    _, nu, ni, nnz, d, n_tags, per, usf = case
    coo = H.make_interactions(nu, ni, nnz, seed=3, ratings=(loss == "logistic"), zipf=0.6)
    item_f = H.tag_features(ni, n_tags, per, seed=11)  # [identity | weighted tags]: few tag columns, all of them hot
    user_f = H.identity_features(nu) if usf == "id" else H.tag_features(nu, 7, 2, seed=13)
    rng = np.random.RandomState(5)
    st = _state(item_f, user_f, d, rng)
    a, b = st.copy(), st.copy()
    shuffle, seeds = H.epoch_inputs(coo, rng)
    options.set(log_samples=loss != "logistic", launches_per_epoch=len(shuffle), warp_kernel=1)
    _run(fast, loss, coo, item_f, user_f, a, shuffle, seeds, coo.data)
    assert options.last_kernel_used == 2 and options.last_plan_flags & 32, (options.last_kernel_used, options.last_plan_flags)
    o = _orc(loss, coo, item_f, user_f, b, shuffle, seeds, coo.data)
    if loss != "logistic":
        neg, sampled = options.last_logs
        assert np.array_equal(sampled, o.sampled) and np.array_equal(neg, o.neg)
    assert options.last_counters == o.counters, (options.last_counters, o.counters)
    tag_rows = slice(ni, ni + n_tags)
    assert not np.array_equal(a.item_embeddings[tag_rows], st.item_embeddings[tag_rows]), "the tag rows were not trained"
    # (a cell updated n times carries up to n half-ulps of the value it had then: logistic updates at EVERY position)
    H.assert_states_within_ulps(a, b, ulps=8, min_exact=0.3)


FROZEN = [
    ("d64-tags", 300, 200, 6000, 64, 12, 3),
    ("d128-c3-like", 400, 300, 8000, 128, 100, 8),
    ("d20-tags", 150, 100, 3000, 20, 12, 3),
]


@pytest.mark.parametrize("case", FROZEN, ids=[c[0] for c in FROZEN])
@pytest.mark.parametrize("loss", ["warp", "bpr"])
def test_frozen_weights_samples_exact_with_the_hot_set(fast, case, loss):
    from lightfm_amd.options import options
    _, nu, ni, nnz, d, n_tags, per = case
    coo = H.make_interactions(nu, ni, nnz, seed=17, zipf=0.6)
    item_f, user_f = H.tag_features(ni, n_tags, per, seed=11), H.identity_features(nu)
    rng = np.random.RandomState(9)
    st = _state(item_f, user_f, d, rng)
    a, b = st.copy(), st.copy()
    zeros = np.zeros_like(coo.data)
    shuffle, seeds = H.epoch_inputs(coo, rng)
    options.set(log_samples=True, ramp_k=-1, warp_kernel=1)
    _run(fast, loss, coo, item_f, user_f, a, shuffle, seeds, zeros)
    assert options.last_kernel_used == 2 and options.last_plan_flags & 32
    o = _orc(loss, coo, item_f, user_f, b, shuffle, seeds, zeros)
    neg, sampled = options.last_logs
    assert np.array_equal(sampled, o.sampled), "sample counts differ"
    assert np.array_equal(neg, o.neg), "negative (rank) indices differ"
    assert options.last_counters == o.counters
    H.assert_states_equal(a, st, exact=True)


@pytest.mark.parametrize("loss", ["bpr", "warp"])
def test_training_with_the_hot_set_learns_like_the_atomic_path(loss):
    from lightfm_amd import LightFM, synthetic
    from lightfm_amd.options import options
    nu, ni = 3000, 2000
    coo = synthetic.make_interactions(nu, ni, 150_000, seed=2)
    feats = synthetic.tag_item_features(ni, n_tags=60, per_item=4)
    rows, cols = np.ascontiguousarray(coo.row), np.ascontiguousarray(coo.col)
    negs = np.random.RandomState(0).randint(0, ni, size=coo.nnz).astype(np.int32)
    acc = {}
    for arm, debug in (("hot", 0), ("atomics", 16384)):
        options.set(debug=debug)
        m = LightFM(no_components=64, loss=loss, random_state=4)
        m.fit(coo, item_features=feats, epochs=4)
        flags = m._last_epoch_stats[-1].get("plan_flags", 0)
        assert bool(flags & 32) == (arm == "hot"), (arm, flags)
        assert np.isfinite(m.item_embeddings).all() and np.isfinite(m.item_biases).all()
        acc[arm] = float(np.mean(m.predict(rows, cols, item_features=feats) > m.predict(rows, negs, item_features=feats)))
    print("pairwise accuracy", acc)
    assert acc["hot"] > 0.8 and abs(acc["hot"] - acc["atomics"]) < 0.02, acc


def test_adagrad_cell_without_root_and_quotient_is_bit_identical():
    """device.hpp: cell_math_adagrad (what the hot-slice kernel evaluates per cell) against cell_math (the float64 cell of
    every other kernel, PYX:416-449) on 2^30 pseudo-random cells at two learning rates: not one differing bit; the exact
    fallback -- results too close to a float32 rounding boundary to be decided without the root -- is taken a few times
    in 10^7."""
    import ctypes as C
    from lightfm_amd import _native as N
    assert N.device_count() > 0
    for lr, seed in ((0.05, 12345), (0.5, 777)):
        bad, slow = C.c_int64(-1), C.c_int64(-1)
        n = 1 << 29
        N.check(N.lib().lfm_selftest_adagrad_cell(C.c_int64(n), C.c_uint32(seed), C.c_float(lr), C.byref(bad), C.byref(slow)))
        print("lr %.2f: %d cells, %d mismatches, %d exact fallbacks" % (lr, n, bad.value, slow.value))
        assert bad.value == 0, bad.value
        assert 0 <= slow.value < n // 10000, slow.value
