"""lightfm_amd.evaluation on the GPU against slow numpy restatements built on model.predict --
the approach of the reference's tests/test_evaluation.py (T_EVAL:34-269)."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fitted():
    from lightfm_amd import LightFM, _native
    assert _native.device_count() > 0
    nu, ni = 50, 40
    full = sp.rand(nu, ni, density=0.3, format="csr", random_state=3)
    full.data[:] = 1.0
    mask = sp.rand(nu, ni, density=0.5, format="csr", random_state=4).astype(bool)
    test = full.multiply(mask).tocsr()
    train = (full - test).tocsr()
    train.eliminate_zeros()
    test.eliminate_zeros()
    model = LightFM(loss="warp", no_components=8, random_state=1).fit(train, epochs=5)
    return model, train.astype(np.float32), test.astype(np.float32)


def _scores(model, nu, ni):
    uids = np.repeat(np.arange(nu, dtype=np.int32), ni)
    iids = np.tile(np.arange(ni, dtype=np.int32), nu)
    return model.predict(uids, iids).reshape(nu, ni)


def _slow_ranks(scores, test, train=None):
    """Pessimistic rank of every test item among all items that are not train positives."""
    out = {}
    for u in range(test.shape[0]):
        s = scores[u].copy()
        excluded = np.zeros(len(s), bool)
        if train is not None:
            excluded[train[u].indices] = True
        for i in test[u].indices:
            others = ~excluded
            others[i] = False
            out[(u, i)] = int(np.sum(s[others] >= s[i]))
    return out


@pytest.mark.parametrize("with_train", [False, True])
def test_precision_recall_reciprocal_rank(fitted, with_train):
    from lightfm_amd.evaluation import precision_at_k, recall_at_k, reciprocal_rank
    model, train, test = fitted
    nu, ni = test.shape
    tr = train if with_train else None
    ranks = _slow_ranks(_scores(model, nu, ni), test, tr)
    k = 5
    prec, rec, mrr = [], [], []
    for u in range(nu):
        items = test[u].indices
        if len(items) == 0:
            continue
        r = np.array([ranks[(u, i)] for i in items])
        prec.append(np.sum(r < k) / k)
        rec.append(np.sum(r < k) / len(items))
        mrr.append(1.0 / (r.min() + 1))
    kw = dict(train_interactions=tr) if with_train else {}
    np.testing.assert_allclose(precision_at_k(model, test, k=k, **kw), prec, rtol=1e-6)
    np.testing.assert_allclose(recall_at_k(model, test, k=k, **kw), rec, rtol=1e-6)
    np.testing.assert_allclose(reciprocal_rank(model, test, **kw), mrr, rtol=1e-6)


def test_auc_matches_pairwise_definition(fitted):
    from lightfm_amd.evaluation import auc_score
    model, train, test = fitted
    nu, ni = test.shape
    scores = _scores(model, nu, ni)
    want = []
    for u in range(nu):
        pos = test[u].indices
        if len(pos) == 0:
            continue
        excluded = np.zeros(ni, bool)
        excluded[train[u].indices] = True
        excluded[pos] = True
        neg = np.where(~excluded)[0]
        if len(neg) == 0:
            want.append(0.5)
            continue
        wins = sum(np.sum(scores[u, p] > scores[u, neg]) for p in pos)
        want.append(wins / (len(pos) * len(neg)))
    got = auc_score(model, test, train_interactions=train)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


def test_intersecting_train_and_test_is_rejected(fitted):
    from lightfm_amd.evaluation import precision_at_k
    model, train, test = fitted
    with pytest.raises(ValueError):
        precision_at_k(model, train, train_interactions=train)
    precision_at_k(model, train, train_interactions=train, check_intersections=False)


# (the default -- unset / "4" -- is the bucket search with its products on the bf16 matrix pipe, split operands)
_RANK_KERNELS = (("bucket-search-bf16", "4"), ("bucket-search", "3"), ("lane-per-user", "2"), ("users-as-rows", "1"), ("scalar", "0"))


def _ranks_by_kernel(model, test, train=None, **kw):
    import os
    out = {}
    try:
        for name, env in _RANK_KERNELS:
            os.environ["LIGHTFM_AMD_RANKS_MFMA"] = env
            out[name] = model.predict_rank(test, train_interactions=train, check_intersections=False, **kw).data
    finally:
        os.environ.pop("LIGHTFM_AMD_RANKS_MFMA", None)
    return out


def test_mfma_prefiltered_ranks_equal_the_scalar_kernel_at_scale():
    """predict_ranks through the three MFMA pre-filters (csrc/predict_kernels.hip: ranks_mfma3_kernel -- the
    default: a lane owns a user, a search over the sorted thresholds counts a score -- ranks_mfma2_kernel, the
    same sweep with a compare chain, and ranks_mfma_kernel) and through the scalar sequential-dot kernel
    on a TRAINED model at the ML-20M item count: every rank identical (an MFMA score only decides
    comparisons outside its rounding band, the rest is re-decided with the sequential dot)."""
    import scipy.sparse as sp
    from lightfm_amd import LightFM, synthetic
    data = synthetic.make_interactions(6000, 26744, 900000, seed=5)
    train, test = synthetic.train_test_split(data, 0.1, seed=1)
    m = LightFM(no_components=64, loss="warp", random_state=1).fit(train, epochs=3)
    ranks = _ranks_by_kernel(m, test, train)
    assert len(ranks["scalar"]) == test.nnz and ranks["scalar"].max() > 100
    assert np.array_equal(ranks["bucket-search-bf16"], ranks["scalar"])
    assert np.array_equal(ranks["bucket-search"], ranks["scalar"])
    assert np.array_equal(ranks["lane-per-user"], ranks["scalar"])
    assert np.array_equal(ranks["users-as-rows"], ranks["scalar"])
    # heavy users (more test items than one pass of the kernels holds) and an odd no_components
    m2 = LightFM(no_components=33, loss="bpr", random_state=2).fit(train, epochs=1)
    heavy = sp.coo_matrix((np.ones(300, np.float32), (np.repeat([3, 4000], 150), np.tile(np.arange(150) * 7, 2))),
                          shape=test.shape, dtype=np.float32)
    out = _ranks_by_kernel(m2, heavy)
    assert np.array_equal(out["bucket-search-bf16"], out["scalar"])
    assert np.array_equal(out["bucket-search"], out["scalar"])
    assert np.array_equal(out["lane-per-user"], out["scalar"])
    assert np.array_equal(out["users-as-rows"], out["scalar"])


@pytest.mark.parametrize("d,n_items", [(8, 37), (32, 1000), (100, 513), (128, 2048)])
def test_mfma_ranks_shapes_and_ties(d, n_items):
    """Every width class of the MFMA sweep (d <= 32, 64, 128), item counts that are not multiples of
    the 32-item tile, items that tie EXACTLY with a test item (duplicated embedding rows: the
    reference counts them, `>=`) and a fresh model whose scores are all within the rounding band."""
    from lightfm_amd import LightFM, synthetic
    data = synthetic.make_interactions(300, n_items, 12 * 300, seed=d)
    train, test = synthetic.train_test_split(data, 0.2, seed=3)
    m = LightFM(no_components=d, loss="warp", random_state=4).fit(train, epochs=2)
    # exact ties: the second half of the items repeats the first half's rows and biases
    h = n_items // 2
    m.item_embeddings[h:2 * h] = m.item_embeddings[:h]
    m.item_biases[h:2 * h] = m.item_biases[:h]
    out = _ranks_by_kernel(m, test, train)
    assert np.array_equal(out["bucket-search-bf16"], out["scalar"])
    assert np.array_equal(out["bucket-search"], out["scalar"])
    assert np.array_equal(out["lane-per-user"], out["scalar"])
    assert np.array_equal(out["users-as-rows"], out["scalar"])
    # the default kernel with the item table cut into three segments (partial counts meet in float atomics; the
    # train-row cursor starts in the middle of a row)
    import os
    os.environ["LIGHTFM_AMD_RANKS_SEGMENTS"] = "3"
    try:
        cut = m.predict_rank(test, train_interactions=train, check_intersections=False).data
    finally:
        os.environ.pop("LIGHTFM_AMD_RANKS_SEGMENTS", None)
    assert np.array_equal(cut, out["scalar"])
    fresh = LightFM(no_components=d, loss="warp", random_state=5)
    fresh._initialize(d, n_items, 300)
    fresh.item_embeddings *= 1e-3
    out = _ranks_by_kernel(fresh, test, train)
    assert np.array_equal(out["bucket-search-bf16"], out["scalar"])
    assert np.array_equal(out["bucket-search"], out["scalar"])
    assert np.array_equal(out["lane-per-user"], out["scalar"])


# ------------------------------------------------ resident scoring session ---

def _fresh_copy(model):
    """A new model object with copies of the weights (no cached device state)."""
    import copy
    from lightfm_amd.lightfm import _WEIGHTS
    other = copy.copy(model)
    other.__dict__.pop("_scoring", None)
    for name in _WEIGHTS:
        setattr(other, name, getattr(model, name).copy())
    return other


def test_scoring_session_is_reused_and_revalidated(fitted):
    """predict / predict_rank keep the embeddings and biases on the device between calls (reference
    call sites LFM:862-870, 979-987 re-wrap the host arrays per call): the same session object serves
    repeated calls, and ANY change of the host arrays -- in place or by assignment -- is seen."""
    model, train, test = fitted
    model = _fresh_copy(model)
    nu, ni = test.shape
    s0 = _scores(model, nu, ni)
    session = model._scoring[0]
    r0 = model.predict_rank(test, train_interactions=train).toarray()
    assert model._scoring[0] is session                     # reused
    np.testing.assert_array_equal(_scores(model, nu, ni), s0)
    # in-place edit of one cell
    model.item_embeddings[3, 2] += 0.5
    s1 = _scores(model, nu, ni)
    np.testing.assert_array_equal(s1, _scores(_fresh_copy(model), nu, ni))
    assert not np.array_equal(s1[:, 3], s0[:, 3]) and np.array_equal(s1[:, 4], s0[:, 4])
    # in-place permutation of rows (same multiset of values)
    model.user_embeddings[[0, 1]] = model.user_embeddings[[1, 0]]
    s2 = _scores(model, nu, ni)
    np.testing.assert_array_equal(s2, _scores(_fresh_copy(model), nu, ni))
    # assignment of a new array
    model.item_biases = model.item_biases + 1.0
    r1 = model.predict_rank(test, train_interactions=train).toarray()
    np.testing.assert_array_equal(r1, _fresh_copy(model).predict_rank(test, train_interactions=train).toarray())
    assert r0.shape == r1.shape


def test_scoring_session_dropped_by_fit_and_pickle(fitted):
    import pickle
    model, train, test = fitted
    model = _fresh_copy(model)
    nu, ni = test.shape
    _scores(model, nu, ni)
    assert "_scoring" in model.__dict__
    clone = pickle.loads(pickle.dumps(model))               # the device handle is not pickled
    assert "_scoring" not in clone.__dict__
    np.testing.assert_array_equal(_scores(clone, nu, ni), _scores(model, nu, ni))
    model.fit_partial(train, epochs=1)
    assert "_scoring" not in model.__dict__                 # stale tables are dropped before training
    np.testing.assert_array_equal(_scores(model, nu, ni), _scores(_fresh_copy(model), nu, ni))


def test_scoring_session_cannot_train(fitted):
    from lightfm_amd import _native as N
    from lightfm_amd._lightfm_fast import CSRMatrix, make_opts
    from lightfm_amd.lightfm import _Session
    model, train, test = fitted
    nu, ni = test.shape
    s = _Session(model._get_lightfm_data(), CSRMatrix(sp.identity(ni, dtype=np.float32, format="csr")),
                 CSRMatrix(sp.identity(nu, dtype=np.float32, format="csr")), scoring=True)
    try:
        coo = train.tocoo()
        s.set_interactions(None, coo.row.astype(np.int32), coo.col.astype(np.int32), coo.data, coo.data)
        s.upload_shuffle(np.arange(coo.nnz, dtype=np.int32))
        opts, _ = make_opts()
        with pytest.raises(ValueError, match="scoring session"):
            s.epoch("warp", 0.0, 0.0, 5, 10, np.array([1], np.uint32), opts)
        with pytest.raises(ValueError):
            s.merge_begin(1)
    finally:
        s.close()


def test_unsorted_train_rows_give_the_same_ranks(fitted):
    """the MFMA sweeps walk each user's train row in column order.  predict_rank sorts a CSR
    that arrives unsorted; the C entry point falls back to the scalar kernel for one that is handed to it."""
    model, train, test = fitted
    ref = model.predict_rank(test, train_interactions=train).toarray()
    rng = np.random.RandomState(0)
    shuffled = train.copy()
    for u in range(shuffled.shape[0]):
        lo, hi = shuffled.indptr[u], shuffled.indptr[u + 1]
        perm = rng.permutation(hi - lo)
        shuffled.indices[lo:hi] = shuffled.indices[lo:hi][perm]
        shuffled.data[lo:hi] = shuffled.data[lo:hi][perm]
    shuffled.has_sorted_indices = False
    np.testing.assert_array_equal(model.predict_rank(test, train_interactions=shuffled).toarray(), ref)


def test_representations_reject_out_of_range_feature_ids(fitted):
    """lfm_session_representations range-checks the feature ids against the embedding table."""
    model, train, test = fitted
    n_feat = model.item_embeddings.shape[0]
    feats = sp.csr_matrix((np.ones(2, np.float32), np.array([0, n_feat - 1], np.int32), np.array([0, 2], np.int32)),
                          shape=(1, n_feat))
    b, e = model.get_item_representations(feats)
    np.testing.assert_allclose(e[0], model.item_embeddings[0] + model.item_embeddings[n_feat - 1], rtol=1e-6)
    bad = sp.csr_matrix((1, n_feat), dtype=np.float32)
    bad.indices, bad.indptr, bad.data = np.array([n_feat + 5], np.int32), np.array([0, 1], np.int32), np.ones(1, np.float32)
    with pytest.raises(ValueError):
        model.get_item_representations(bad)


def test_concurrent_predict_with_different_feature_matrices(fitted):
    """The reference's predict is read-only on the model; threads may share it (joblib-threaded
    evaluation, serving).  The cached scoring session is state (its feature matrices are swapped per
    call, ctypes drops the GIL): a per-model lock serialises its use and a caller that finds it busy
    scores on a one-shot session.  8 threads x 30 calls with per-thread feature matrices must give
    exactly the single-threaded scores."""
    import threading
    model, train, test = fitted
    model = _fresh_copy(model)
    nu, ni = test.shape
    uids = np.repeat(np.arange(nu, dtype=np.int32), ni)
    iids = np.tile(np.arange(ni, dtype=np.int32), nu)
    n_feat = model.item_embeddings.shape[0]
    mats = []
    for t in range(8):
        m = sp.random(ni, n_feat, density=0.2, format="csr", random_state=100 + t, dtype=np.float32)
        mats.append((m + sp.identity(ni, dtype=np.float32, format="csr")[:, :n_feat]).tocsr().astype(np.float32))
    want = [model.predict(uids, iids, item_features=m) for m in mats]
    ranks_want = model.predict_rank(test, train_interactions=train).toarray()
    errors = []

    def worker(t):
        try:
            for _ in range(30):
                got = model.predict(uids, iids, item_features=mats[t])
                if not np.array_equal(got, want[t]):
                    errors.append("thread %d: scores differ" % t)
                    return
                if t == 0 and not np.array_equal(model.predict_rank(test, train_interactions=train).toarray(), ranks_want):
                    errors.append("ranks differ")
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]


def test_resident_train_matrix_is_revalidated_between_calls():
    """predict_ranks keeps the train matrix of the last call on the scoring session (csrc/session.hip: train_keep) and
    re-validates it with a position-sensitive checksum of its index arrays: the same matrix again, the same matrix with
    two entries of a row swapped for other columns IN PLACE, a different matrix of the same shape and nnz, and no train
    matrix at all must each give the ranks of a session that has never seen another one."""
    from lightfm_amd import LightFM, synthetic
    nu, ni = 3000, 2500
    data = synthetic.make_interactions(nu, ni, 260_000, seed=9)
    train, test = synthetic.split_off_test(data, data.nnz - 20_000, seed=3)
    train_a = train.tocsr().astype(np.float32)
    train_a.sort_indices()
    assert train_a.nnz >= (1 << 16), "the matrix must be large enough to be kept"
    test = test.tocsr().astype(np.float32)
    model = LightFM(loss="warp", no_components=32, random_state=2).fit(train, epochs=2)

    def fresh(tr):
        return _fresh_copy(model).predict_rank(test, train_interactions=tr).toarray()

    r_a = model.predict_rank(test, train_interactions=train_a).toarray()
    np.testing.assert_array_equal(r_a, fresh(train_a))
    np.testing.assert_array_equal(model.predict_rank(test, train_interactions=train_a).toarray(), r_a)  # kept and reused
    # in place: user 5 loses two train positives and gains two others (same nnz, same indptr)
    train_b = train_a.copy()
    lo, hi = train_b.indptr[5], train_b.indptr[5 + 1]
    assert hi - lo >= 4
    free = np.setdiff1d(np.arange(ni), np.union1d(train_b.indices[lo:hi], test[5].indices))[:2]
    row = np.sort(np.concatenate([train_b.indices[lo + 2:hi], free])).astype(np.int32)
    train_b.indices[lo:hi] = row
    r_b = model.predict_rank(test, train_interactions=train_b).toarray()
    np.testing.assert_array_equal(r_b, fresh(train_b))
    assert not np.array_equal(r_b[5], r_a[5]) and np.array_equal(r_b[6], r_a[6])
    # the same arrays edited IN PLACE between two calls
    train_b.indices[lo:hi] = train_a.indices[lo:hi]
    np.testing.assert_array_equal(model.predict_rank(test, train_interactions=train_b).toarray(), r_a)
    # no train matrix
    np.testing.assert_array_equal(model.predict_rank(test).toarray(), _fresh_copy(model).predict_rank(test).toarray())


@pytest.mark.parametrize("d", [3, 10, 32, 33, 64, 100, 128])
def test_bf16_split_scores_stay_inside_their_rounding_band(d):
    """predict_ranks' default sweep takes its products from the bf16 matrix pipe (two-way split operands) and re-decides every
    comparison inside a rounding band with the reference's sequential dot; the band (csrc/predict_kernels.hip:
    ranks_bf_kappa_t / _s) bounds the split's truncation and the accumulations, whose internal rounding the ISA leaves
    unspecified (taken as faithful).  The device self-test runs 2 x 10^5 random 32 x 32 tiles per setting through the sweep's own
    instruction sequence: no pair may leave its band, and the largest fraction of the band actually used is printed
    (components of equal magnitude and spread over 6 / 14 binades)."""
    import ctypes as C
    from lightfm_amd import _native as N
    assert N.device_count() > 0
    for spread in (1, 6, 14):
        worst, beyond = C.c_float(-1.0), C.c_int64(-1)
        N.check(N.lib().lfm_selftest_ranks_bf16_band(C.c_int64(200_000), C.c_uint32(1234 + d), C.c_int32(d), C.c_int32(spread),
                                                     C.byref(worst), C.byref(beyond)))
        print("d %d, %d binades: largest |bf16-split score - sequential dot| = %.3f of the band, %d pairs beyond it" % (d, spread, worst.value, beyond.value))
        assert beyond.value == 0 and 0.0 < worst.value < 1.0, (d, spread, worst.value, beyond.value)
