"""GPU parity tests proper: the HIP path, called through the C ABI's Python
binding (lightfm_amd._lightfm_fast, the drop-in for the reference's native
module), against the CPU oracle on the same seeded inputs.

Bars:
  * serial mode (one wavefront, reference order, reference rand_r streams):
    weights, biases, accumulators BIT-EXACT and (negative, sampled) logs exact
    for every loss / schedule / feature layout / alpha;
  * parallel mode with frozen weights (sample_weight = 0, the reference's own
    trick, tests/test_movielens.py:517-533): (negative, sampled) per position and
    the draw / probe totals exact, weights untouched;
  * parallel mode, conflict-free inputs: weights bit-exact as well;
  * predict / predict_ranks / auc / in_positives: exact.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle
from tests import helpers as H
from tests.test_oracle_vs_reference import LOSS_CASES, _problem, _run_orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fast():
    import lightfm_amd._lightfm_fast as f
    from lightfm_amd import _native
    assert _native.device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    return f


@pytest.fixture(autouse=True)
def _reset_options():
    from lightfm_amd.options import options
    options.set(mode="parallel", launches_per_epoch=0, first_batch=0, max_waves=0, log_samples=False,
                warp_kernel=0)
    yield
    options.set(mode="parallel", launches_per_epoch=0, first_batch=0, max_waves=0, log_samples=False,
                warp_kernel=0)


def _hip_struct(fast, st):
    return fast.FastLightFM(*st.arrays(), st.d, int(st.schedule == "adadelta"), st.lr, st.rho,
                            st.eps, st.max_sampled)


def _run_hip(fast, loss, coo, item_f, user_f, st, shuffle, seeds, alpha, k=3, n=5, weight=None):
    Cm = fast.CSRMatrix
    fl = _hip_struct(fast, st)
    pos = H.positives_csr(coo)
    w = weight if weight is not None else (coo.data if loss != "logistic" else np.ones_like(coo.data))
    rs = H.FixedRandom(seeds)
    if loss == "warp":
        fast.fit_warp(Cm(item_f), Cm(user_f), Cm(pos), coo.row, coo.col, coo.data, w, shuffle, fl,
                      0.05, alpha, alpha * 2, len(seeds), rs)
    elif loss == "bpr":
        fast.fit_bpr(Cm(item_f), Cm(user_f), Cm(pos), coo.row, coo.col, coo.data, w, shuffle, fl,
                     0.05, alpha, alpha * 2, len(seeds), rs)
    elif loss == "warp-kos":
        fast.fit_warp_kos(Cm(item_f), Cm(user_f), Cm(pos), coo.row, shuffle, fl, 0.05, alpha,
                          alpha * 2, k, n, len(seeds), rs)
    else:
        fast.fit_logistic(Cm(item_f), Cm(user_f), coo.row, coo.col, coo.data, w, shuffle, fl, 0.05,
                          alpha, alpha * 2, 1)


def _orc_logged(loss, coo, item_f, user_f, st, shuffle, seeds, alpha, rng_mode, weight=None,
                k=3, n=5):
    o = oracle.Opts(len(shuffle), rng_mode=rng_mode, log=True)
    pos = H.positives_csr(coo)
    w = weight if weight is not None else coo.data
    if loss == "warp":
        oracle.fit_warp(item_f, user_f, pos, coo.row, coo.col, coo.data, w, shuffle, st, alpha,
                        alpha * 2, seeds, o)
    elif loss == "bpr":
        oracle.fit_bpr(item_f, user_f, pos, coo.row, coo.col, coo.data, w, shuffle, st, alpha,
                       alpha * 2, seeds, o)
    else:
        oracle.fit_warp_kos(item_f, user_f, pos, coo.row, shuffle, st, alpha, alpha * 2, k, n,
                            seeds, o)
    return o


@pytest.mark.parametrize("loss", ["warp", "bpr", "logistic", "warp-kos"])
@pytest.mark.parametrize("case", LOSS_CASES, ids=[c[0] for c in LOSS_CASES])
def test_serial_mode_bit_exact(fast, loss, case):
    from lightfm_amd.options import options
    options.set(mode="serial", log_samples=True)
    coo, item_f, user_f, st, rng, alpha = _problem(case)
    a, b = st.copy(), st.copy()
    for _ in range(2):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _run_hip(fast, loss, coo, item_f, user_f, a, shuffle, seeds, alpha)
        if loss == "logistic":
            _run_orc(loss, coo, item_f, user_f, b, shuffle, seeds, alpha)
        else:
            o = _orc_logged(loss, coo, item_f, user_f, b, shuffle, seeds, alpha, rng_mode=0)
            neg, sampled = options.last_logs
            assert np.array_equal(sampled, o.sampled), "sample counts differ"
            assert np.array_equal(neg, o.neg), "negative (rank) indices differ"
            assert options.last_counters == o.counters
    assert not np.array_equal(a.item_embeddings, st.item_embeddings)
    if loss in ("warp", "warp-kos"):
        H.assert_states_equal(a, b, exact=True)
    else:
        # exp() runs in the device's libm: float32 results agree to the last bit in
        # practice; the stated tolerance is 1e-6 relative (north star allows 1e-4)
        H.assert_states_equal(a, b, exact=False, rtol=1e-6, atol=1e-9)


def test_serial_mode_multi_stream(fast):
    """n_seeds > 1: the reference's static chunks, run back to back."""
    from lightfm_amd.options import options
    options.set(mode="serial")
    coo, item_f, user_f, st, rng, alpha = _problem(LOSS_CASES[2])
    a, b = st.copy(), st.copy()
    shuffle, seeds = H.epoch_inputs(coo, rng, num_threads=3)
    _run_hip(fast, "warp", coo, item_f, user_f, a, shuffle, seeds, 0.0)
    _run_orc("warp", coo, item_f, user_f, b, shuffle, seeds, 0.0)
    H.assert_states_equal(a, b, exact=True)


@pytest.mark.parametrize("loss", ["warp", "bpr", "warp-kos"])
@pytest.mark.parametrize("case", [LOSS_CASES[0], LOSS_CASES[1], LOSS_CASES[2]],
                         ids=[c[0] for c in LOSS_CASES[:3]])
@pytest.mark.parametrize("first_batch", [1, 4])
def test_parallel_frozen_weights_samples_exact(fast, loss, case, first_batch):
    """sample_weight = 0 => every loss is 0 => weights never change, so each position's
    (negative, sampled) depends only on its own PRNG stream: order independent."""
    from lightfm_amd.options import options
    options.set(mode="parallel", log_samples=True, launches_per_epoch=3, first_batch=first_batch)
    coo, item_f, user_f, st, rng, _ = _problem(case)
    if loss == "warp-kos":
        pytest.skip("k-OS ignores sample weights (PYX:1039): weights cannot be frozen")
    zeros = np.zeros_like(coo.data)
    a, b = st.copy(), st.copy()
    shuffle, seeds = H.epoch_inputs(coo, rng)
    _run_hip(fast, loss, coo, item_f, user_f, a, shuffle, seeds, 0.0, weight=zeros)
    o = _orc_logged(loss, coo, item_f, user_f, b, shuffle, seeds, 0.0, rng_mode=1, weight=zeros)
    neg, sampled = options.last_logs
    assert np.array_equal(sampled, o.sampled)
    assert np.array_equal(neg, o.neg)
    assert options.last_counters == o.counters
    H.assert_states_equal(a, st, exact=True)
    H.assert_states_equal(a, b, exact=True)


def _conflict_free(n_users, n_items, seed):
    """Interactions in which every user and every item appears at most once."""
    rng = np.random.RandomState(seed)
    n = min(n_users, n_items)
    u = rng.permutation(n_users)[:n].astype(np.int32)
    i = rng.permutation(n_items)[:n].astype(np.int32)
    data = rng.choice([-1.0, 1.0], size=n).astype(np.float32)
    return sp.coo_matrix((data, (u, i)), shape=(n_users, n_items), dtype=np.float32)


@pytest.mark.parametrize("d", [8, 64, 100])
def test_parallel_conflict_free_matches_sequential(fast, d):
    """With no two interactions sharing a row, Hogwild has no races: the atomic read-modify-write
    path must reproduce the sequential result.  The bar is 1e-6 relative, not bit equality: the
    logistic loss evaluates exp() in the device libm (the oracle uses the host's) and an update
    published as old + fl32(new - old) differs from `new` where the subtraction is inexact; the
    bit-exact statements of this suite are the serial-mode tests above and the WARP kernels'
    one-interaction-per-launch / disjoint-group tests (tests/test_hip_warp_tile.py)."""
    coo = _conflict_free(300, 280, 5)
    item_f, user_f = H.identity_features(280), H.identity_features(300)
    rng = np.random.RandomState(1)
    st = oracle.State(280, 300, d, rng)
    a, b = st.copy(), st.copy()
    for _ in range(2):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _run_hip(fast, "logistic", coo, item_f, user_f, a, shuffle, seeds, 0.0)
        _run_orc("logistic", coo, item_f, user_f, b, shuffle, seeds, 0.0)
    H.assert_states_equal(a, b, exact=False, rtol=1e-6, atol=1e-9)


def test_parallel_training_learns_like_the_oracle(fast):
    """Full parallel training is not order-deterministic (neither is the reference with
    num_threads > 1); check the fit quality statistically: mean positive-vs-random score
    margin within a few percent of the sequential oracle's after 5 epochs."""
    coo = H.make_interactions(400, 300, 12000, seed=21)
    item_f, user_f = H.identity_features(300), H.identity_features(400)
    rng = np.random.RandomState(5)
    st = oracle.State(300, 400, 32, rng)
    a, b = st.copy(), st.copy()
    for _ in range(5):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _run_hip(fast, "warp", coo, item_f, user_f, a, shuffle, seeds, 0.0)
        _run_orc("warp", coo, item_f, user_f, b, shuffle, seeds, 0.0)

    def margin(s):
        r = np.random.RandomState(0)
        pos = oracle.predict(item_f, user_f, coo.row, coo.col, s)
        neg = oracle.predict(item_f, user_f, coo.row,
                             r.randint(0, 300, size=len(coo.row)).astype(np.int32), s)
        return float(np.mean(pos - neg)), float(np.mean(pos > neg))

    (ma, aa), (mb, ab) = margin(a), margin(b)
    assert ab > 0.8
    assert abs(aa - ab) < 0.04, (aa, ab)  # Hogwild: run-to-run variation
    assert abs(ma - mb) / abs(mb) < 0.2, (ma, mb)


@pytest.mark.parametrize("case", [LOSS_CASES[1], LOSS_CASES[3]], ids=["id-d33", "tags-both"])
def test_predict_ranks_auc_exact(fast, case):
    coo, item_f, user_f, st, rng, _ = _problem(case)
    shuffle, seeds = H.epoch_inputs(coo, rng)
    _run_orc("warp", coo, item_f, user_f, st, shuffle, seeds, 0.0)
    nu, ni = coo.shape
    uids = np.repeat(np.arange(nu, dtype=np.int32), ni)
    iids = np.tile(np.arange(ni, dtype=np.int32), nu)
    want = oracle.predict(item_f, user_f, uids, iids, st)
    got = np.empty_like(want)
    Cm = fast.CSRMatrix
    fast.predict_lightfm(Cm(item_f), Cm(user_f), uids, iids, got, _hip_struct(fast, st), 1)
    assert np.array_equal(want, got)

    train = H.positives_csr(coo).astype(np.float32)
    test = H.make_interactions(nu, ni, 300, seed=99).tocsr().astype(np.float32)
    test = (test - test.multiply(train.astype(bool))).tocsr().astype(np.float32)
    test.eliminate_zeros()
    test.sort_indices()
    r_orc = np.zeros_like(test.data)
    r_hip = np.zeros_like(test.data)
    oracle.predict_ranks(item_f, user_f, test, train, r_orc, st)
    fast.predict_ranks(Cm(item_f), Cm(user_f), Cm(test), Cm(train), r_hip, _hip_struct(fast, st), 1)
    assert np.array_equal(r_orc, r_hip)
    assert r_orc.max() > 0

    ranks_a = sp.csr_matrix((r_orc.copy(), test.indices, test.indptr), shape=test.shape)
    ranks_b = sp.csr_matrix((r_orc.copy(), test.indices, test.indptr), shape=test.shape)
    ntp = np.squeeze(np.array(train.getnnz(axis=1)).astype(np.int32))
    auc_a = np.zeros(nu, np.float32)
    auc_b = np.zeros(nu, np.float32)
    oracle.auc_from_rank(ranks_a, ntp, ranks_a.data, auc_a)
    fast.calculate_auc_from_rank(Cm(ranks_b), ntp, ranks_b.data, auc_b, 1)
    assert np.array_equal(auc_a, auc_b)
    assert np.array_equal(ranks_a.data, ranks_b.data)


# ------------------------------------------------------------------------------------------ models wider than 512 components
# (the reference has no bound on no_components, PYX:185-259; fit_kernels_wide.hip: sixteen coordinates per lane, d <= 1 024)

WIDE_CASES = [
    ("id-adagrad-d1001", 30, 24, 260, 1001, None, None, "adagrad", 0.0, True),
    ("tags-both-adadelta-alpha-d600", 24, 30, 220, 600, "tagsnorm", "tags", "adadelta", 1e-4, False),
    ("id-adagrad-d1024", 20, 16, 150, 1024, None, None, "adagrad", 0.0, False),
]


@pytest.mark.parametrize("loss", ["warp", "bpr", "logistic", "warp-kos"])
@pytest.mark.parametrize("case", WIDE_CASES, ids=[c[0] for c in WIDE_CASES])
def test_wide_models_serial_mode_bit_exact(fast, loss, case):
    from lightfm_amd.options import options
    options.set(mode="serial", log_samples=True)
    coo, item_f, user_f, st, rng, alpha = _problem(case)
    a, b = st.copy(), st.copy()
    for _ in range(2):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _run_hip(fast, loss, coo, item_f, user_f, a, shuffle, seeds, alpha)
        assert options.last_kernel_used == 0
        if loss == "logistic":
            _run_orc(loss, coo, item_f, user_f, b, shuffle, seeds, alpha)
        else:
            o = _orc_logged(loss, coo, item_f, user_f, b, shuffle, seeds, alpha, rng_mode=0)
            neg, sampled = options.last_logs
            assert np.array_equal(sampled, o.sampled) and np.array_equal(neg, o.neg)
            assert options.last_counters == o.counters
    assert not np.array_equal(a.item_embeddings, st.item_embeddings)
    if loss in ("warp", "warp-kos"):
        H.assert_states_equal(a, b, exact=True)
    else:
        H.assert_states_equal(a, b, exact=False, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("loss", ["warp", "bpr", "warp-kos"])
def test_wide_models_parallel_mode(fast, loss):
    """d = 1 001 in the shipped mode: one interaction per launch is the sequential run (negatives and counters exact, arrays
    within the bar of float-atomic publication); then predict / predict_ranks on the trained model equal the oracle's."""
    from lightfm_amd.options import options
    coo, item_f, user_f, st, rng, alpha = _problem(WIDE_CASES[0])
    options.set(mode="parallel", log_samples=True, launches_per_epoch=coo.nnz)
    a, b = st.copy(), st.copy()
    shuffle, seeds = H.epoch_inputs(coo, rng)
    _run_hip(fast, loss, coo, item_f, user_f, a, shuffle, seeds, alpha)
    o = _orc_logged(loss, coo, item_f, user_f, b, shuffle, seeds, alpha, rng_mode=1)
    neg, sampled = options.last_logs
    assert np.array_equal(sampled, o.sampled) and np.array_equal(neg, o.neg) and options.last_counters == o.counters
    H.assert_states_equal(a, b, exact=False, rtol=2e-5, atol=2e-6)
    nu, ni = coo.shape
    uids = np.repeat(np.arange(nu, dtype=np.int32), ni)
    iids = np.tile(np.arange(ni, dtype=np.int32), nu)
    want = oracle.predict(item_f, user_f, uids, iids, b)
    got = np.empty_like(want)
    Cm = fast.CSRMatrix
    fast.predict_lightfm(Cm(item_f), Cm(user_f), uids, iids, got, _hip_struct(fast, b), 1)
    assert np.array_equal(want, got)
    train = H.positives_csr(coo).astype(np.float32)
    test = H.make_interactions(nu, ni, 150, seed=99).tocsr().astype(np.float32)
    test = (test - test.multiply(train.astype(bool))).tocsr().astype(np.float32)
    test.eliminate_zeros()
    test.sort_indices()
    r_orc, r_hip = np.zeros_like(test.data), np.zeros_like(test.data)
    oracle.predict_ranks(item_f, user_f, test, train, r_orc, b)
    fast.predict_ranks(Cm(item_f), Cm(user_f), Cm(test), Cm(train), r_hip, _hip_struct(fast, b), 1)
    assert np.array_equal(r_orc, r_hip) and r_orc.max() > 0


def test_in_positives_truth_table(fast):
    # reference tests/test_fast_functions.py:9-17
    mat = sp.csr_matrix(np.array([[0, 1], [1, 0]], dtype=np.float32))
    fn = getattr(fast, "__test_in_positives")
    for r in range(2):
        for c in range(2):
            assert fn(r, c, fast.CSRMatrix(mat)) == bool(mat[r, c])
    # long rows exercise the 64-ary search rounds
    rng = np.random.RandomState(0)
    cols = np.sort(rng.choice(200000, size=9000, replace=False)).astype(np.int32)
    big = sp.csr_matrix((np.ones(9000, np.float32), cols, np.array([0, 9000], np.int32)),
                        shape=(1, 200000))
    present = set(cols.tolist())
    for c in list(cols[::997]) + [0, 1, 199999, 12345, int(cols[0]), int(cols[-1])]:
        assert fn(0, int(c), fast.CSRMatrix(big)) == (int(c) in present)


def _golden_files():
    import glob
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return sorted(glob.glob(os.path.join(gold, "*.npz")))


@pytest.mark.parametrize("path", _golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_hip_reproduces_the_reference_fixtures(fast, path):
    """The HIP path against the golden vectors the REFERENCE produced (tests/golden/, generated
    by make_golden.py from the reference's own compiled extension): two serial-mode epochs give
    the 12 weight arrays, then predictions, ranks and AUCs -- no oracle in between."""
    import os
    from lightfm_amd.options import options
    case_id, loss = os.path.basename(path)[:-4].split("__")
    case = {c[0]: c for c in LOSS_CASES}[case_id]
    gold = np.load(path)
    coo, item_f, user_f, st, rng, alpha = _problem(case)
    options.set(mode="serial")
    for _ in range(2):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _run_hip(fast, loss, coo, item_f, user_f, st, shuffle, seeds, alpha)
    options.set(mode="parallel")
    exact = loss in ("warp", "warp-kos")
    for n in oracle.ARRAYS:
        if exact:
            assert np.array_equal(getattr(st, n), gold[n]), n
        else:  # exp() of the device libm: last-bit differences, stated tolerance 1e-6
            np.testing.assert_allclose(getattr(st, n), gold[n], rtol=1e-6, atol=1e-9, err_msg=n)
    if not exact:
        for n in oracle.ARRAYS:  # score the reference's own weights from here on
            getattr(st, n)[...] = gold[n]
    nu, ni = coo.shape
    Cm = fast.CSRMatrix
    uids = np.repeat(np.arange(nu, dtype=np.int32), ni)
    iids = np.tile(np.arange(ni, dtype=np.int32), nu)
    pred = np.empty(len(uids), np.float32)
    fast.predict_lightfm(Cm(item_f), Cm(user_f), uids, iids, pred, _hip_struct(fast, st), 1)
    assert np.array_equal(pred[:: max(1, len(pred) // 512)], gold["predictions"])
    train, test = H.rank_problem(coo)
    ranks = np.zeros_like(test.data)
    fast.predict_ranks(Cm(item_f), Cm(user_f), Cm(test), Cm(train), ranks, _hip_struct(fast, st), 1)
    assert np.array_equal(ranks, gold["ranks"])
    rmat = sp.csr_matrix((ranks.copy(), test.indices, test.indptr), shape=test.shape)
    ntp = np.squeeze(np.array(train.getnnz(axis=1)).astype(np.int32))
    auc = np.zeros(nu, np.float32)
    fast.calculate_auc_from_rank(Cm(rmat), ntp, rmat.data, auc, 1)
    assert np.array_equal(auc, gold["auc"])
