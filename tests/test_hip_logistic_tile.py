"""GPU parity tests of the logistic lane-group kernel (lightfm_amd/csrc/logistic_tile.hip): fit_logistic (PYX:694-781) of a
narrow identity model -- the reference's literal default, LightFM(): logistic loss, no_components = 10 -- on rows that hold W, G, b
and bG of a feature in one 128-byte line.

Bars:
  * one interaction per launch: the kernel is then sequential, and two epochs later all arrays equal the oracle's within the bar of
    float-atomic publication (old + float32(new - old)); counters exact;
  * concurrent launches whose interactions share no row (eight per wavefront pass, several wavefronts, several passes): the
    sequential oracle's result again -- the lane exchanges, the hand-over of the accumulator deltas, the LDS transposition and the
    line-wide publication under load;
  * outside its scope (d = 16, an L2 penalty, feature matrices) the row-stream kernel runs as before;
  * full-concurrency training learns what the row-stream kernel learns.
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


def _hip(fast, coo, st, shuffle, weight, item_alpha=0.0):
    Cm = fast.CSRMatrix
    fl = fast.FastLightFM(*st.arrays(), st.d, 0, st.lr, st.rho, st.eps, st.max_sampled)
    item_f, user_f = H.identity_features(coo.shape[1]), H.identity_features(coo.shape[0])
    fast.fit_logistic(Cm(item_f), Cm(user_f), coo.row, coo.col, coo.data, weight, shuffle, fl, 0.05, item_alpha, 0.0, 1)


def _orc(coo, st, shuffle, weight):
    o = oracle.Opts(len(shuffle), rng_mode=1, log=True)
    item_f, user_f = H.identity_features(coo.shape[1]), H.identity_features(coo.shape[0])
    oracle.fit_logistic(item_f, user_f, coo.row, coo.col, coo.data, weight, shuffle, st, 0.0, 0.0, o)
    return o


def _state(ni, nu, d, seed):
    rng = np.random.RandomState(seed)
    st = oracle.State(ni, nu, d, rng, max_sampled=10)
    st.item_embeddings *= 4 * d
    st.user_embeddings *= 4 * d
    st.item_biases[:] = rng.randn(ni).astype(np.float32) * 0.3
    st.user_biases[:] = rng.randn(nu).astype(np.float32) * 0.3
    return st


def _labels(coo, rng):
    """+-1 labels with a few zeros (PYX:751-755: y <= 0 is the label 0) and non-trivial sample weights"""
    y = np.where(rng.rand(coo.nnz) < 0.5, 1.0, -1.0).astype(np.float32)
    y[rng.rand(coo.nnz) < 0.05] = 0.0
    w = (0.25 + rng.rand(coo.nnz) * 1.5).astype(np.float32)
    return sp.coo_matrix((y, (coo.row, coo.col)), shape=coo.shape, dtype=np.float32), w


@pytest.mark.parametrize("d", [4, 8, 10, 12, 3])
def test_one_interaction_per_launch_matches_the_oracle(fast, d):
    from lightfm_amd.options import options
    nu, ni = 40, 30
    rng = np.random.RandomState(4)
    coo, weight = _labels(H.make_interactions(nu, ni, 260, seed=3), rng)
    st = _state(ni, nu, d, 7)
    a, b = st.copy(), st.copy()
    options.set(launches_per_epoch=coo.nnz, update_mode=0)
    for _ in range(2):
        shuffle, _ = H.epoch_inputs(coo, rng)
        _hip(fast, coo, a, shuffle, weight)
        o = _orc(coo, b, shuffle, weight)
        assert options.last_kernel_used == 1 and options.last_plan_flags & 256, (options.last_kernel_used, options.last_plan_flags)
        assert options.last_counters == o.counters
    assert not np.array_equal(a.item_embeddings, st.item_embeddings) and not np.array_equal(a.user_biases, st.user_biases)
    # (the float adder of the atomic unit does not round like v_add_f32; every cell is updated a dozen times)
    H.assert_states_equal(a, b, exact=False, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("d", [4, 10, 12])
@pytest.mark.parametrize("per_launch", [8, 64, 200], ids=["one-pass", "eight-waves", "tail-inside-a-pass"])
def test_concurrent_conflict_free_launches_match_the_oracle(fast, d, per_launch):
    """No two interactions of a launch share a user or an item: whatever the concurrency, the result is the sequential one."""
    from lightfm_amd.options import options
    n_launches = 5
    n = per_launch * n_launches
    nu, ni = 2 * per_launch + 7, 2 * per_launch + 11
    rng = np.random.RandomState(11 + d)
    rows = np.concatenate([rng.permutation(nu)[:per_launch] for _ in range(n_launches)]).astype(np.int32)
    cols = np.concatenate([rng.permutation(ni)[:per_launch] for _ in range(n_launches)]).astype(np.int32)
    coo, weight = _labels(sp.coo_matrix((np.ones(n, np.float32), (rows, cols)), shape=(nu, ni), dtype=np.float32), rng)
    coo = sp.coo_matrix((coo.data, (rows, cols)), shape=(nu, ni), dtype=np.float32)  # (keeps the order and the duplicates)
    shuffle = np.arange(n, dtype=np.int32)  # launch l covers positions [l * per_launch, (l + 1) * per_launch)
    st = _state(ni, nu, d, 5)
    a, b = st.copy(), st.copy()
    options.set(launches_per_epoch=n_launches, update_mode=0, ramp_k=-1)
    _hip(fast, coo, a, shuffle, weight)
    assert options.last_kernel_used == 1 and options.last_plan_flags & 256
    o = _orc(coo, b, shuffle, weight)
    assert options.last_counters == o.counters
    H.assert_states_within_ulps(a, b, ulps=4)


def test_outside_its_scope_other_kernels_run(fast, monkeypatch):
    from lightfm_amd.options import options
    rng = np.random.RandomState(2)
    coo, weight = _labels(H.make_interactions(60, 50, 500, seed=9), rng)
    shuffle, _ = H.epoch_inputs(coo, rng)
    # (d = 16: the tile kernel's logistic instantiation, csrc/warp_tile_bpr.hip -- plan_flags bit 11; tests/test_hip_bpr_tile.py)
    # (... and with an L2 penalty its regularised one, at the default width too)
    for d, alpha, env, used in ((16, 0.0, None, 1), (10, 1e-6, None, 1), (10, 0.0, "0", 2)):
        if env is not None:
            monkeypatch.setenv("LIGHTFM_AMD_LOGISTIC_TILE", env)
        st = _state(50, 60, d, 1)
        _hip(fast, coo, st, shuffle, weight, item_alpha=alpha)
        assert options.last_kernel_used == used and not (options.last_plan_flags & 256), (d, alpha, env, options.last_plan_flags)
        assert bool(options.last_plan_flags & 2048) == (used == 1)


def test_training_learns_like_the_row_stream_kernel(monkeypatch):
    """The reference's default model at full concurrency, through this kernel and through the row-stream kernel
    (LIGHTFM_AMD_LOGISTIC_TILE=0): the same accuracy on the training labels."""
    from lightfm_amd import LightFM
    nu, ni = 14000, 11000
    pos = H.make_interactions(nu, ni, 300_000, seed=12, zipf=0.8)
    rng = np.random.RandomState(0)
    neg_r, neg_c = rng.randint(0, nu, size=pos.nnz).astype(np.int32), rng.randint(0, ni, size=pos.nnz).astype(np.int32)
    order = rng.permutation(2 * pos.nnz)
    data = sp.coo_matrix((np.concatenate([np.ones(pos.nnz, np.float32), -np.ones(pos.nnz, np.float32)])[order],
                          (np.concatenate([pos.row, neg_r])[order], np.concatenate([pos.col, neg_c])[order])), shape=(nu, ni), dtype=np.float32)
    acc = {}
    for arm, env in (("tile", "1"), ("row-stream", "0")):
        monkeypatch.setenv("LIGHTFM_AMD_LOGISTIC_TILE", env)
        m = LightFM(random_state=7)  # logistic, no_components = 10
        m.fit(data, epochs=8)
        st = m._last_epoch_stats[-1]
        assert bool(st["plan_flags"] & 256) == (arm == "tile") and st["kernel_used"] == (1 if arm == "tile" else 2), (arm, st)
        p = m.predict(np.ascontiguousarray(data.row), np.ascontiguousarray(data.col))
        acc[arm] = float(np.mean((p > 0) == (data.data > 0)))
    print("accuracy on the training labels", acc)
    assert acc["tile"] > 0.7 and abs(acc["tile"] - acc["row-stream"]) < 0.01, acc


# ------------------------------------------------------------------------------------------------ fit_bpr_tile_kernel

def _hip_bpr(fast, coo, st, shuffle, seeds, weight):
    Cm = fast.CSRMatrix
    fl = fast.FastLightFM(*st.arrays(), st.d, 0, st.lr, st.rho, st.eps, st.max_sampled)
    item_f, user_f = H.identity_features(coo.shape[1]), H.identity_features(coo.shape[0])
    fast.fit_bpr(Cm(item_f), Cm(user_f), Cm(H.positives_csr(coo)), coo.row, coo.col, coo.data, weight, shuffle, fl, 0.05, 0.0, 0.0,
                 len(seeds), H.FixedRandom(seeds))


def _orc_bpr(coo, st, shuffle, seeds, weight):
    o = oracle.Opts(len(shuffle), rng_mode=1, log=True)
    item_f, user_f = H.identity_features(coo.shape[1]), H.identity_features(coo.shape[0])
    oracle.fit_bpr(item_f, user_f, H.positives_csr(coo), coo.row, coo.col, coo.data, weight, shuffle, st, 0.0, 0.0, seeds, o)
    return o


@pytest.mark.parametrize("d", [4, 10, 12])
def test_bpr_one_interaction_per_launch_matches_the_oracle(fast, d):
    """Dense rows (a third of the catalogue per user): the first candidate is a positive in a third of the interactions, both
    are in a ninth (the one-draw-at-a-time path with its late line fetch).  Sequential launches: negatives, draw counts and
    counters exact, the arrays within the bar of float-atomic publication."""
    from lightfm_amd.options import options
    nu, ni = 24, 60
    rng = np.random.RandomState(8 + d)
    dense = rng.rand(nu, ni) < 0.33
    dense[:, 0] = True
    m = sp.coo_matrix(dense.astype(np.float32))
    vals = (1.0 + rng.rand(m.nnz)).astype(np.float32)
    vals[rng.rand(m.nnz) < 0.1] = 0.0  # (PYX:1116-1117: not a positive, skipped before any draw)
    coo = sp.coo_matrix((vals, (m.row.astype(np.int32), m.col.astype(np.int32))), shape=(nu, ni), dtype=np.float32)
    st = _state(ni, nu, d, 3)
    a, b = st.copy(), st.copy()
    options.set(log_samples=True, launches_per_epoch=coo.nnz, update_mode=0)
    multi = 0
    for _ in range(2):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _hip_bpr(fast, coo, a, shuffle, seeds, coo.data)
        assert options.last_kernel_used == 1 and options.last_plan_flags & 512, (options.last_kernel_used, options.last_plan_flags)
        o = _orc_bpr(coo, b, shuffle, seeds, coo.data)
        neg, sampled = options.last_logs
        assert np.array_equal(sampled, o.sampled), int((sampled != o.sampled).sum())
        assert np.array_equal(neg, o.neg), int((neg != o.neg).sum())
        assert options.last_counters == o.counters
        multi += int((o.sampled > 2).sum())
    assert multi > 20, "the path past the two speculative candidates is not exercised"
    assert not np.array_equal(a.item_embeddings, st.item_embeddings) and not np.array_equal(a.item_biases, st.item_biases)
    H.assert_states_equal(a, b, exact=False, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("d", [4, 12])
def test_bpr_user_with_the_whole_catalogue(fast, d):
    """Every draw is one of the user's positives, and the last of the no_examples draws may be the positive itself (PYX:1123-1127:
    the loop ends on its bound): the reference then updates ONE row twice in sequence, first as the positive, then as the negative."""
    from lightfm_amd.options import options
    nu, ni = 2, 7
    rng = np.random.RandomState(d)
    coo = sp.coo_matrix(np.ones((nu, ni), dtype=np.float32))
    coo = sp.coo_matrix(((1.0 + rng.rand(coo.nnz)).astype(np.float32), (coo.row.astype(np.int32), coo.col.astype(np.int32))), shape=(nu, ni))
    st = _state(ni, nu, d, 5)
    a, b = st.copy(), st.copy()
    options.set(log_samples=True, launches_per_epoch=coo.nnz, update_mode=0)
    same = 0
    for _ in range(3):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _hip_bpr(fast, coo, a, shuffle, seeds, coo.data)
        assert options.last_plan_flags & 512
        o = _orc_bpr(coo, b, shuffle, seeds, coo.data)
        neg, sampled = options.last_logs
        assert np.array_equal(sampled, o.sampled) and np.array_equal(neg, o.neg) and (o.sampled == coo.nnz).all()
        same += int((o.neg == coo.col[shuffle]).sum())
    assert same >= 3, "no interaction drew its own positive last"
    H.assert_states_equal(a, b, exact=False, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("d", [4, 10, 12])
@pytest.mark.parametrize("waves", [0, 8], ids=["full-grid", "two-workgroups-many-passes"])
def test_bpr_frozen_weights_samples_exact(fast, d, waves):
    """sample_weight = 0 freezes the model (the reference's own trick): across the whole grid and with two workgroups walking
    hundreds of passes -- records three passes deep, candidates requested two passes ahead, launch tails inside a pass -- every
    position's negative and draw count and the counters equal the oracle's, and no array moves."""
    from lightfm_amd.options import options
    nu, ni = 3000, 500
    coo = H.make_interactions(nu, ni, 60_011, seed=29, ratings=True, zipf=0.7)
    rng = np.random.RandomState(4)
    st = _state(ni, nu, d, 6)
    a, b = st.copy(), st.copy()
    zeros = np.zeros_like(coo.data)
    shuffle, seeds = H.epoch_inputs(coo, rng)
    options.set(log_samples=True, launches_per_epoch=3, ramp_k=-1, max_waves=waves)
    _hip_bpr(fast, coo, a, shuffle, seeds, zeros)
    assert options.last_kernel_used == 1 and options.last_plan_flags & 512
    o = _orc_bpr(coo, b, shuffle, seeds, zeros)
    neg, sampled = options.last_logs
    assert np.array_equal(sampled, o.sampled), "draw counts differ at %d positions" % int((sampled != o.sampled).sum())
    assert np.array_equal(neg, o.neg), "negatives differ at %d positions" % int((neg != o.neg).sum())
    assert options.last_counters == o.counters
    assert (o.sampled > 2).sum() > 50
    H.assert_states_equal(a, st, exact=True)


def test_bpr_training_learns_like_the_row_stream_kernel(monkeypatch):
    from lightfm_amd import LightFM
    coo = H.make_interactions(14000, 11000, 450_000, seed=12, zipf=0.8)
    rows, cols = np.ascontiguousarray(coo.row), np.ascontiguousarray(coo.col)
    negs = np.random.RandomState(0).randint(0, 11000, size=coo.nnz).astype(np.int32)
    acc = {}
    for arm, env in (("tile", "1"), ("row-stream", "0")):
        monkeypatch.setenv("LIGHTFM_AMD_BPR_TILE", env)
        m = LightFM(loss="bpr", random_state=7)  # no_components = 10
        m.fit(coo, epochs=6)
        st = m._last_epoch_stats[-1]
        assert bool(st["plan_flags"] & 512) == (arm == "tile") and st["kernel_used"] == (1 if arm == "tile" else 2), (arm, st)
        acc[arm] = float(np.mean(m.predict(rows, cols) > m.predict(rows, negs)))
    print("pairwise accuracy", acc)
    assert acc["tile"] > 0.6 and abs(acc["tile"] - acc["row-stream"]) < 0.01, acc  # (both arms: 0.645 after six epochs)
