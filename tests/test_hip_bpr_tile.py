"""GPU parity tests of fit_bpr (PYX:1074-1182) and fit_logistic (PYX:694-781) on the BPR / logistic instantiations of the
lane-group tile kernel (lightfm_amd/csrc/warp_tile_bpr.hip; warp_tile_kernel.hpp, LOSS = LFM_LOSS_BPR_ID / LFM_LOSS_LOGISTIC_ID):
identity models wider than the narrow lane-group kernels of logistic_tile.hip take -- 12 < d <= 64 at four interactions per wavefront pass (rows by LDS-DMA), d <= 128
at two, d <= 256 at one.

Bars (those of the WARP tile kernels):
  * one interaction per launch: the kernel is then sequential -- every negative, draw count and counter equals the oracle's, and
    two epochs later all arrays equal it within the bar of float-atomic publication (old + float32(new - old));
  * a user with the whole catalogue: every draw is a positive, the loop ends on its bound and the last draw may be the positive
    itself -- the reference then updates ONE row twice in sequence;
  * frozen weights (sample_weight = 0) at full concurrency and with two workgroups walking hundreds of passes: negatives, draw
    counts, counters exact, no array moves -- the batches after the first (both speculative candidates were positives) included;
  * outside its scope (adadelta, feature matrices) the row-stream kernel runs as before; with an L2 penalty the REG instantiations run:
    one interaction per launch equals the sequential oracle incl. the two scales (the fold tests: tests/test_hip_round2.py);
  * full-concurrency training learns what the row-stream kernel learns;
  * logistic: one interaction per launch and conflict-free concurrent launches (every user and item once: no two interactions
    share a row) equal the sequential oracle; both labels, zero values, sample weights.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu

_DEFAULTS = dict(mode="parallel", launches_per_epoch=0, first_batch=0, max_waves=0, log_samples=False,
                 warp_kernel=0, feat_kernel=0, update_mode=0, debug=0, ramp_k=0, shared_cap=0)
WIDTHS = [16, 30, 64, 100, 200]  # device rows of 16 / 32 / 64 (four per pass), 100 (two per pass), 200 floats (one per pass)
BPR_TILE, LGT_TILE = 1024, 2048   # lfm_opts.plan_flags bits 10, 11


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


def _state(ni, nu, d, seed, schedule="adagrad"):
    rng = np.random.RandomState(seed)
    st = oracle.State(ni, nu, d, rng, schedule=schedule, max_sampled=10)
    st.item_embeddings *= 4 * np.sqrt(d)
    st.user_embeddings *= 4 * np.sqrt(d)
    st.item_biases[:] = rng.randn(ni).astype(np.float32) * 0.3
    st.user_biases[:] = rng.randn(nu).astype(np.float32) * 0.3
    return st


def _hip(fast, coo, st, shuffle, seeds, weight, item_f=None, user_f=None, alpha=0.0):
    Cm = fast.CSRMatrix
    fl = fast.FastLightFM(*st.arrays(), st.d, int(st.schedule == "adadelta"), st.lr, st.rho, st.eps, st.max_sampled)
    item_f = H.identity_features(coo.shape[1]) if item_f is None else item_f
    user_f = H.identity_features(coo.shape[0]) if user_f is None else user_f
    fast.fit_bpr(Cm(item_f), Cm(user_f), Cm(H.positives_csr(coo)), coo.row, coo.col, coo.data, weight, shuffle, fl, 0.05, alpha, alpha,
                 len(seeds), H.FixedRandom(seeds))


def _orc(coo, st, shuffle, seeds, weight):
    o = oracle.Opts(len(shuffle), rng_mode=1, log=True)
    item_f, user_f = H.identity_features(coo.shape[1]), H.identity_features(coo.shape[0])
    oracle.fit_bpr(item_f, user_f, H.positives_csr(coo), coo.row, coo.col, coo.data, weight, shuffle, st, 0.0, 0.0, seeds, o)
    return o


@pytest.mark.parametrize("d", WIDTHS)
def test_one_interaction_per_launch_matches_the_oracle(fast, d):
    """Dense rows (a third of the catalogue per user): the first candidate is a positive in a third of the interactions, both of
    the first batch are in a ninth (the later batches of two)."""
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
        _hip(fast, coo, a, shuffle, seeds, coo.data)
        assert options.last_kernel_used == 1 and options.last_plan_flags & BPR_TILE, (options.last_kernel_used, options.last_plan_flags)
        o = _orc(coo, b, shuffle, seeds, coo.data)
        neg, sampled = options.last_logs
        assert np.array_equal(sampled, o.sampled), int((sampled != o.sampled).sum())
        assert np.array_equal(neg, o.neg), int((neg != o.neg).sum())
        assert options.last_counters == o.counters
        multi += int((o.sampled > 2).sum())
    assert multi > 20, "the batches after the first are not exercised"
    assert not np.array_equal(a.item_embeddings, st.item_embeddings) and not np.array_equal(a.item_biases, st.item_biases)
    H.assert_states_equal(a, b, exact=False, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("d", [16, 64, 100, 200])
def test_user_with_the_whole_catalogue(fast, d):
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
        _hip(fast, coo, a, shuffle, seeds, coo.data)
        assert options.last_plan_flags & BPR_TILE
        o = _orc(coo, b, shuffle, seeds, coo.data)
        neg, sampled = options.last_logs
        assert np.array_equal(sampled, o.sampled) and np.array_equal(neg, o.neg) and (o.sampled == coo.nnz).all()
        assert options.last_counters == o.counters
        same += int((o.neg == coo.col[shuffle]).sum())
    assert same >= 3, "no interaction drew its own positive last"
    H.assert_states_equal(a, b, exact=False, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("d", WIDTHS)
@pytest.mark.parametrize("waves", [0, 8], ids=["full-grid", "two-workgroups-many-passes"])
def test_frozen_weights_samples_exact(fast, d, waves):
    from lightfm_amd.options import options
    nu, ni = 3000, 500
    coo = H.make_interactions(nu, ni, 60_011, seed=29, ratings=True, zipf=0.7)
    rng = np.random.RandomState(4)
    st = _state(ni, nu, d, 6)
    a, b = st.copy(), st.copy()
    zeros = np.zeros_like(coo.data)
    shuffle, seeds = H.epoch_inputs(coo, rng)
    options.set(log_samples=True, launches_per_epoch=3, ramp_k=-1, max_waves=waves)
    _hip(fast, coo, a, shuffle, seeds, zeros)
    assert options.last_kernel_used == 1 and options.last_plan_flags & BPR_TILE
    o = _orc(coo, b, shuffle, seeds, zeros)
    neg, sampled = options.last_logs
    assert np.array_equal(sampled, o.sampled), "draw counts differ at %d positions" % int((sampled != o.sampled).sum())
    assert np.array_equal(neg, o.neg), "negatives differ at %d positions" % int((neg != o.neg).sum())
    assert options.last_counters == o.counters
    assert (o.sampled > 2).sum() > 50
    H.assert_states_equal(a, st, exact=True)


def test_conflict_free_concurrent_launch_matches_the_oracle(fast):
    """Four interactions per launch = ONE pass of one wavefront with all four lane groups busy.  Users and positives are all
    distinct; BPR's negatives come from the interaction list, so item rows may still coincide -- negatives and draw counts are
    exact everywhere, and the USER rows of positions whose two item rows no earlier position touched equal the oracle's."""
    from lightfm_amd.options import options
    n = 48
    rng = np.random.RandomState(2)
    coo = sp.coo_matrix(((1.0 + rng.rand(n)).astype(np.float32), (rng.permutation(n).astype(np.int32), rng.permutation(n).astype(np.int32))),
                        shape=(n, n), dtype=np.float32)
    st = _state(n, n, 64, 9)
    a, b = st.copy(), st.copy()
    shuffle, seeds = H.epoch_inputs(coo, rng)
    options.set(log_samples=True, launches_per_epoch=n // 4, update_mode=0)
    _hip(fast, coo, a, shuffle, seeds, coo.data)
    assert options.last_plan_flags & BPR_TILE
    o = _orc(coo, b, shuffle, seeds, coo.data)
    neg, sampled = options.last_logs
    assert np.array_equal(neg, o.neg) and np.array_equal(sampled, o.sampled)
    touched, clean = set(), []
    for k, row in enumerate(shuffle):
        p_, n_ = int(coo.col[row]), int(o.neg[k])
        if p_ not in touched and n_ not in touched:
            clean.append(int(coo.row[row]))
        touched.update((p_, n_))
    assert len(clean) > 10
    np.testing.assert_allclose(a.user_embeddings[clean], b.user_embeddings[clean], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(a.user_biases[clean], b.user_biases[clean], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("what", ["adadelta", "item-features", "disabled"])
def test_outside_its_scope_the_row_stream_kernel_runs(fast, what, monkeypatch):
    from lightfm_amd.options import options
    coo = H.make_interactions(300, 200, 5000, seed=3)
    rng = np.random.RandomState(1)
    item_f = H.tag_features(200, 12, 3, 11) if what == "item-features" else None
    st = _state(200 if item_f is None else item_f.shape[1], 300, 32, 2, schedule="adadelta" if what == "adadelta" else "adagrad")
    shuffle, seeds = H.epoch_inputs(coo, rng)
    if what == "disabled":
        monkeypatch.setenv("LIGHTFM_AMD_BPR_WIDE_TILE", "0")
    _hip(fast, coo, st, shuffle, seeds, coo.data, item_f=item_f)
    assert options.last_kernel_used == 2 and not options.last_plan_flags & BPR_TILE, (options.last_kernel_used, options.last_plan_flags)


def test_training_learns_like_the_row_stream_kernel(monkeypatch):
    from lightfm_amd import LightFM
    coo = H.make_interactions(14000, 11000, 450_000, seed=12, zipf=0.8)
    rows, cols = np.ascontiguousarray(coo.row), np.ascontiguousarray(coo.col)
    negs = np.random.RandomState(0).randint(0, 11000, size=coo.nnz).astype(np.int32)
    acc = {}
    for arm, env in (("tile", "1"), ("row-stream", "0")):
        monkeypatch.setenv("LIGHTFM_AMD_BPR_WIDE_TILE", env)
        m = LightFM(loss="bpr", no_components=64, random_state=7)
        m.fit(coo, epochs=6)
        st = m._last_epoch_stats[-1]
        assert bool(st["plan_flags"] & BPR_TILE) == (arm == "tile") and st["kernel_used"] == (1 if arm == "tile" else 2), (arm, st)
        acc[arm] = float(np.mean(m.predict(rows, cols) > m.predict(rows, negs)))
    print("pairwise accuracy", acc)
    assert acc["tile"] > 0.6 and abs(acc["tile"] - acc["row-stream"]) < 0.01, acc


# ------------------------------------------------------------------------------------------------ fit_logistic

def _labels(coo, rng):
    """+-1 labels with a few zeros (PYX:751-755: y <= 0 is the label 0) and non-trivial sample weights"""
    y = np.where(rng.rand(coo.nnz) < 0.5, 1.0, -1.0).astype(np.float32)
    y[rng.rand(coo.nnz) < 0.05] = 0.0
    w = (0.25 + rng.rand(coo.nnz) * 1.5).astype(np.float32)
    return sp.coo_matrix((y, (coo.row, coo.col)), shape=coo.shape, dtype=np.float32), w


def _hip_logistic(fast, coo, st, shuffle, weight):
    Cm = fast.CSRMatrix
    fl = fast.FastLightFM(*st.arrays(), st.d, 0, st.lr, st.rho, st.eps, st.max_sampled)
    item_f, user_f = H.identity_features(coo.shape[1]), H.identity_features(coo.shape[0])
    fast.fit_logistic(Cm(item_f), Cm(user_f), coo.row, coo.col, coo.data, weight, shuffle, fl, 0.05, 0.0, 0.0, 1)


def _orc_logistic(coo, st, shuffle, weight):
    o = oracle.Opts(len(shuffle), rng_mode=1, log=True)
    item_f, user_f = H.identity_features(coo.shape[1]), H.identity_features(coo.shape[0])
    oracle.fit_logistic(item_f, user_f, coo.row, coo.col, coo.data, weight, shuffle, st, 0.0, 0.0, o)
    return o


@pytest.mark.parametrize("d", WIDTHS)
def test_logistic_one_interaction_per_launch_matches_the_oracle(fast, d):
    from lightfm_amd.options import options
    rng = np.random.RandomState(d)
    coo, w = _labels(H.make_interactions(40, 30, 400, seed=5), rng)
    st = _state(30, 40, d, 2)
    a, b = st.copy(), st.copy()
    options.set(launches_per_epoch=coo.nnz, update_mode=0)
    for _ in range(2):
        shuffle, _ = H.epoch_inputs(coo, rng)
        _hip_logistic(fast, coo, a, shuffle, w)
        assert options.last_kernel_used == 1 and options.last_plan_flags & LGT_TILE, (options.last_kernel_used, options.last_plan_flags)
        o = _orc_logistic(coo, b, shuffle, w)
        assert options.last_counters == o.counters
    assert not np.array_equal(a.item_embeddings, st.item_embeddings) and not np.array_equal(a.user_biases, st.user_biases)
    H.assert_states_equal(a, b, exact=False, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("d", WIDTHS)
@pytest.mark.parametrize("waves", [0, 8], ids=["full-grid", "two-workgroups-many-passes"])
def test_logistic_conflict_free_concurrent_launches_match_the_oracle(fast, d, waves):
    """Every user and every item at most once: Hogwild has no races, so all groups of all wavefronts of ONE launch (and two
    workgroups walking many passes) must reproduce the sequential result."""
    from lightfm_amd.options import options
    n = 3000
    rng = np.random.RandomState(3)
    base = sp.coo_matrix((np.ones(n, np.float32), (rng.permutation(n).astype(np.int32), rng.permutation(n + 40)[:n].astype(np.int32))),
                         shape=(n, n + 40), dtype=np.float32)
    coo, w = _labels(base, rng)
    st = _state(n + 40, n, d, 4)
    a, b = st.copy(), st.copy()
    options.set(launches_per_epoch=1, ramp_k=-1, max_waves=waves, update_mode=0)
    shuffle, _ = H.epoch_inputs(coo, rng)
    _hip_logistic(fast, coo, a, shuffle, w)
    assert options.last_kernel_used == 1 and options.last_plan_flags & LGT_TILE
    o = _orc_logistic(coo, b, shuffle, w)
    assert options.last_counters == o.counters
    H.assert_states_equal(a, b, exact=False, rtol=2e-5, atol=2e-6)


def test_logistic_training_learns_like_the_row_stream_kernel(monkeypatch):
    from lightfm_amd import LightFM
    rng = np.random.RandomState(1)
    base = H.make_interactions(14000, 11000, 600_000, seed=12, zipf=0.8)
    # labels a rank-4 model generates: learnable, both classes
    U, V = rng.randn(14000, 4), rng.randn(11000, 4)
    y = np.where((U[base.row] * V[base.col]).sum(1) + 0.3 * rng.randn(base.nnz) > 0, 1.0, -1.0).astype(np.float32)
    data = sp.coo_matrix((y, (base.row, base.col)), shape=base.shape, dtype=np.float32)
    acc = {}
    for arm, env in (("tile", "1"), ("row-stream", "0")):
        monkeypatch.setenv("LIGHTFM_AMD_BPR_WIDE_TILE", env)
        m = LightFM(loss="logistic", no_components=32, random_state=7)
        m.fit(data, epochs=8)
        st = m._last_epoch_stats[-1]
        assert bool(st["plan_flags"] & LGT_TILE) == (arm == "tile") and st["kernel_used"] == (1 if arm == "tile" else 2), (arm, st)
        p = m.predict(np.ascontiguousarray(data.row), np.ascontiguousarray(data.col))
        acc[arm] = float(np.mean((p > 0) == (data.data > 0)))
    print("accuracy on the training labels", acc)
    assert acc["tile"] > 0.7 and abs(acc["tile"] - acc["row-stream"]) < 0.01, acc


# ------------------------------------------------------------------------------------------------ lazy L2 regularisation

@pytest.mark.parametrize("loss", ["bpr", "logistic"])
@pytest.mark.parametrize("d", [16, 64, 100])
def test_regularised_one_interaction_per_launch_matches_the_oracle(fast, loss, d):
    """item_alpha, user_alpha != 0 (PYX:640-691): the REG instantiations -- scaled representations in the scoring and in the
    gradients, cells multiplied by 1 + alpha lr, the scales advanced by the interaction's average learning rate (2 (d + 1) cells
    for logistic, 3 (d + 1) for BPR).  Sequential launches: the arrays equal the oracle's (an epoch call ends with the
    scales folded into them, PYX:775-781, 1176-1182)."""
    from lightfm_amd.options import options
    rng = np.random.RandomState(d)
    base = H.make_interactions(40, 30, 400, seed=5)
    alpha = 2e-3
    st = _state(30, 40, d, 2)
    a, b = st.copy(), st.copy()
    Cm = fast.CSRMatrix
    item_f, user_f = H.identity_features(30), H.identity_features(40)
    options.set(launches_per_epoch=base.nnz, update_mode=0, log_samples=(loss == "bpr"))
    for _ in range(2):
        if loss == "bpr":
            coo, w = base, base.data
        else:
            coo, w = _labels(base, rng)
        shuffle, seeds = H.epoch_inputs(coo, rng)
        fl = fast.FastLightFM(*a.arrays(), a.d, 0, a.lr, a.rho, a.eps, a.max_sampled)
        if loss == "bpr":
            fast.fit_bpr(Cm(item_f), Cm(user_f), Cm(H.positives_csr(coo)), coo.row, coo.col, coo.data, w, shuffle, fl, 0.05, alpha, 2 * alpha,
                         len(seeds), H.FixedRandom(seeds))
            o = oracle.Opts(len(shuffle), rng_mode=1, log=True)
            oracle.fit_bpr(item_f, user_f, H.positives_csr(coo), coo.row, coo.col, coo.data, w, shuffle, b, alpha, 2 * alpha, seeds, o)
            neg, sampled = options.last_logs
            assert np.array_equal(neg, o.neg) and np.array_equal(sampled, o.sampled)
        else:
            fast.fit_logistic(Cm(item_f), Cm(user_f), coo.row, coo.col, coo.data, w, shuffle, fl, 0.05, alpha, 2 * alpha, 1)
            o = oracle.Opts(len(shuffle), rng_mode=1, log=True)
            oracle.fit_logistic(item_f, user_f, coo.row, coo.col, coo.data, w, shuffle, b, alpha, 2 * alpha, o)
        assert options.last_kernel_used == 1 and options.last_plan_flags & (BPR_TILE if loss == "bpr" else LGT_TILE)
        assert options.last_counters == o.counters
    H.assert_states_equal(a, b, exact=False, rtol=5e-5, atol=5e-6)


# ------------------------------------------------------------------------------------------------ randomised

def test_randomised_sequential_parity_of_the_identity_kernels():
    """tools/identity_fuzz.py: 150 random problems (1..40 users, 2..60 items, densities up to the whole catalogue, widths 1..256 incl.
    padded ones, labels with zeros, sample weights, with and without an L2 penalty) through fit_bpr / fit_logistic, one interaction
    per launch, against the oracle: all four identity kernels (lane-group d <= 12, tile instantiations above) take part."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "identity_fuzz.py"), "150", "11"], capture_output=True, text=True, timeout=600)
    tail = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-2000:]
    assert out.returncode == 0 and "150 cases, 0 failed" in tail, (out.stdout[-3000:], out.stderr[-2000:])
    for bit in ("512", "1024", "256", "2048"):
        assert ("1, %s)" % bit) in tail, tail


@pytest.mark.parametrize("loss", ["warp", "bpr", "logistic"])
def test_plain_store_user_rows_only_above_48_components(loss):
    """lfm_opts.user_store (csrc/session.hip "rare_collisions"): the user row of an update is written by plain stores only for
    no_components > 48 -- at or below that the switch buys no kernel time and a lost user update costs more
    (profiles/r06_narrow_quality20m.txt, r06_ustore_ab.txt).  A problem the collision-rate rule itself accepts: 200 000
    users with ~2.5 interactions each (12 288 in flight x sum c_u^2 / n^2 ~ 0.09 <= 0.3), tables in uncached memory."""
    from lightfm_amd import LightFM
    rng = np.random.RandomState(5)
    nu, ni, n = 200_000, 5_000, 500_000
    y = np.ones(n, np.float32) if loss != "logistic" else np.where(rng.rand(n) < 0.5, 1.0, -1.0).astype(np.float32)
    coo = sp.coo_matrix((y, (rng.randint(0, nu, n).astype(np.int32), rng.randint(0, ni, n).astype(np.int32))), shape=(nu, ni), dtype=np.float32)
    took = {}
    for d in (32, 48, 64):
        m = LightFM(loss=loss, no_components=d, random_state=3)
        m.fit(coo, epochs=2)
        st = m._last_epoch_stats[-1]
        assert st["kernel_used"] == 1, (d, st)
        took[d] = st["user_store"]
    assert took == {32: 0, 48: 0, 64: 1}, took
