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


_RANK_KERNELS = (("lane-per-user", "2"), ("users-as-rows", "1"), ("scalar", "0"))


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
    """predict_ranks through both MFMA pre-filters (csrc/predict_kernels.hip: ranks_mfma2_kernel -- the
    default, a lane owns a user -- and ranks_mfma_kernel) and through the scalar sequential-dot kernel
    on a TRAINED model at the ML-20M item count: every rank identical (an MFMA score only decides
    comparisons outside its rounding band, the rest is re-decided with the sequential dot)."""
    import scipy.sparse as sp
    from lightfm_amd import LightFM, synthetic
    data = synthetic.make_interactions(6000, 26744, 900000, seed=5)
    train, test = synthetic.train_test_split(data, 0.1, seed=1)
    m = LightFM(no_components=64, loss="warp", random_state=1).fit(train, epochs=3)
    ranks = _ranks_by_kernel(m, test, train)
    assert len(ranks["scalar"]) == test.nnz and ranks["scalar"].max() > 100
    assert np.array_equal(ranks["lane-per-user"], ranks["scalar"])
    assert np.array_equal(ranks["users-as-rows"], ranks["scalar"])
    # heavy users (more test items than one pass of the kernels holds) and an odd no_components
    m2 = LightFM(no_components=33, loss="bpr", random_state=2).fit(train, epochs=1)
    heavy = sp.coo_matrix((np.ones(300, np.float32), (np.repeat([3, 4000], 150), np.tile(np.arange(150) * 7, 2))),
                          shape=test.shape, dtype=np.float32)
    out = _ranks_by_kernel(m2, heavy)
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
    assert np.array_equal(out["lane-per-user"], out["scalar"])
    assert np.array_equal(out["users-as-rows"], out["scalar"])
    fresh = LightFM(no_components=d, loss="warp", random_state=5)
    fresh._initialize(d, n_items, 300)
    fresh.item_embeddings *= 1e-3
    out = _ranks_by_kernel(fresh, test, train)
    assert np.array_equal(out["lane-per-user"], out["scalar"])
