"""Pins oracle/lfm_oracle.c to the reference itself: every function is compared
bit-for-bit with the reference's shipped Cython output compiled by
`make -C oracle ref` (strict / LIGHTFM_NO_CFLAGS build).  CPU only.

Skipped when oracle/_ref is absent (no /root/reference and no prebuilt .so);
tests/test_golden.py then pins the oracle against committed fixtures instead.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle
from tests import helpers as H

LOSS_CASES = [
    # (name, n_users, n_items, nnz, d, item_feats, user_feats, schedule, alpha, ratings)
    ("id-adagrad", 60, 40, 500, 8, None, None, "adagrad", 0.0, False),
    ("id-adagrad-d33", 50, 70, 600, 33, None, None, "adagrad", 0.0, True),
    ("tags-adagrad", 60, 40, 500, 8, "tags", None, "adagrad", 0.0, False),
    ("tags-both-adadelta", 40, 50, 400, 6, "tags", "tags", "adadelta", 0.0, True),
    ("id-alpha", 60, 40, 500, 8, None, None, "adagrad", 1e-3, False),
    ("tags-alpha-adadelta", 40, 50, 400, 5, "tagsnorm", "tags", "adadelta", 1e-4, False),
]


def _problem(case):
    name, nu, ni, nnz, d, itf, usf, sched, alpha, ratings = case
    coo = H.make_interactions(nu, ni, nnz, seed=7, ratings=ratings)
    item_f = H.identity_features(ni) if itf is None else H.tag_features(
        ni, 12, 3, 11, normalise=(itf == "tagsnorm"))
    user_f = H.identity_features(nu) if usf is None else H.tag_features(nu, 7, 2, 13)
    rng = np.random.RandomState(3)
    st = oracle.State(item_f.shape[1], user_f.shape[1], d, rng, schedule=sched, max_sampled=7)
    return coo, item_f, user_f, st, rng, alpha


def _run_ref(ref, loss, coo, item_f, user_f, st, shuffle, seeds, alpha, k=3, n=5):
    C = ref.CSRMatrix
    fl = st.ref_struct(ref)
    pos = H.positives_csr(coo)
    w = coo.data if loss != "logistic" else np.ones_like(coo.data)
    if loss == "warp":
        ref.fit_warp(C(item_f), C(user_f), C(pos), coo.row, coo.col, coo.data, w, shuffle, fl,
                     0.05, alpha, alpha * 2, len(seeds), H.FixedRandom(seeds))
    elif loss == "bpr":
        ref.fit_bpr(C(item_f), C(user_f), C(pos), coo.row, coo.col, coo.data, w, shuffle, fl,
                    0.05, alpha, alpha * 2, len(seeds), H.FixedRandom(seeds))
    elif loss == "warp-kos":
        ref.fit_warp_kos(C(item_f), C(user_f), C(pos), coo.row, shuffle, fl, 0.05, alpha,
                         alpha * 2, k, n, len(seeds), H.FixedRandom(seeds))
    else:
        ref.fit_logistic(C(item_f), C(user_f), coo.row, coo.col, coo.data, w, shuffle, fl, 0.05,
                         alpha, alpha * 2, 1)


def _run_orc(loss, coo, item_f, user_f, st, shuffle, seeds, alpha, k=3, n=5):
    pos = H.positives_csr(coo)
    w = coo.data if loss != "logistic" else np.ones_like(coo.data)
    if loss == "warp":
        return oracle.fit_warp(item_f, user_f, pos, coo.row, coo.col, coo.data, w, shuffle, st,
                               alpha, alpha * 2, seeds)
    if loss == "bpr":
        return oracle.fit_bpr(item_f, user_f, pos, coo.row, coo.col, coo.data, w, shuffle, st,
                              alpha, alpha * 2, seeds)
    if loss == "warp-kos":
        return oracle.fit_warp_kos(item_f, user_f, pos, coo.row, shuffle, st, alpha, alpha * 2, k,
                                   n, seeds)
    return oracle.fit_logistic(item_f, user_f, coo.row, coo.col, coo.data, w, shuffle, st, alpha,
                               alpha * 2)


@pytest.mark.parametrize("loss", ["warp", "bpr", "logistic", "warp-kos"])
@pytest.mark.parametrize("case", LOSS_CASES, ids=[c[0] for c in LOSS_CASES])
def test_fit_bit_exact_vs_reference(ref_strict, loss, case):
    coo, item_f, user_f, st, rng, alpha = _problem(case)
    a, b = st.copy(), st.copy()
    for _ in range(3):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _run_ref(ref_strict, loss, coo, item_f, user_f, a, shuffle, seeds, alpha)
        _run_orc(loss, coo, item_f, user_f, b, shuffle, seeds, alpha)
    assert not np.array_equal(a.item_embeddings, st.item_embeddings)
    H.assert_states_equal(a, b, exact=True)


def test_rand_r_stream_matches_reference_sampling(ref_strict):
    """rand_r is not exported by the reference; pin it through fit_warp's choice of
    negatives: with max_sampled=1, lr huge on G so only the chosen item's G moves."""
    coo, item_f, user_f, st, rng, _ = _problem(LOSS_CASES[0])
    st.max_sampled = 1
    a, b = st.copy(), st.copy()
    shuffle, seeds = H.epoch_inputs(coo, rng)
    _run_ref(ref_strict, "warp", coo, item_f, user_f, a, shuffle, seeds, 0.0)
    o = oracle.Opts(len(shuffle), log=True)
    pos = H.positives_csr(coo)
    oracle.fit_warp(item_f, user_f, pos, coo.row, coo.col, coo.data, coo.data, shuffle, b, 0.0,
                    0.0, seeds, o)
    H.assert_states_equal(a, b)
    touched = np.unique(o.neg[o.neg >= 0])
    moved = np.where((a.item_bias_gradients != 1).ravel())[0]
    assert set(touched).issubset(set(moved))


def test_predict_and_ranks_and_auc(ref_strict):
    coo, item_f, user_f, st, rng, _ = _problem(LOSS_CASES[2])
    shuffle, seeds = H.epoch_inputs(coo, rng)
    _run_orc("warp", coo, item_f, user_f, st, shuffle, seeds, 0.0)
    C = ref_strict.CSRMatrix
    nu, ni = coo.shape
    uids = np.repeat(np.arange(nu, dtype=np.int32), ni)
    iids = np.tile(np.arange(ni, dtype=np.int32), nu)
    want = np.empty(len(uids), np.float32)
    ref_strict.predict_lightfm(C(item_f), C(user_f), uids, iids, want, st.ref_struct(ref_strict), 1)
    got = oracle.predict(item_f, user_f, uids, iids, st)
    assert np.array_equal(want, got)

    train = H.positives_csr(coo).astype(np.float32)
    test = H.make_interactions(nu, ni, 300, seed=99).tocsr().astype(np.float32)
    test = (test - test.multiply(train.astype(bool))).tocsr().astype(np.float32)
    test.eliminate_zeros()
    test.sort_indices()
    r_ref = np.zeros_like(test.data)
    r_orc = np.zeros_like(test.data)
    ref_strict.predict_ranks(C(item_f), C(user_f), C(test), C(train), r_ref,
                             st.ref_struct(ref_strict), 1)
    oracle.predict_ranks(item_f, user_f, test, train, r_orc, st)
    assert np.array_equal(r_ref, r_orc)
    assert r_ref.max() > 0

    ranks = sp.csr_matrix((r_ref.copy(), test.indices, test.indptr), shape=test.shape)
    ranks2 = sp.csr_matrix((r_ref.copy(), test.indices, test.indptr), shape=test.shape)
    ntp = np.squeeze(np.array(train.getnnz(axis=1)).astype(np.int32))
    auc_ref = np.zeros(nu, np.float32)
    auc_orc = np.zeros(nu, np.float32)
    ref_strict.calculate_auc_from_rank(C(ranks), ntp, ranks.data, auc_ref, 1)
    oracle.auc_from_rank(ranks2, ntp, ranks2.data, auc_orc)
    assert np.array_equal(auc_ref, auc_orc)


def test_in_positives_truth_table(ref_strict):
    # same table as the reference's tests/test_fast_functions.py:9-17
    mat = sp.csr_matrix(np.array([[0, 1], [1, 0]], dtype=np.float32))
    for r in range(2):
        for c in range(2):
            assert oracle.in_positives(c, r, mat) == bool(mat[r, c])
            fn = getattr(ref_strict, "__test_in_positives", None)
            if fn is not None:
                assert fn(r, c, ref_strict.CSRMatrix(mat)) == bool(mat[r, c])
