"""Drop-in contract of `lightfm_amd.LightFM` on the GPU: the behaviours the reference pins in its
own tests/test_api.py (T_API) and tests/test_movielens.py (T_ML), restated against this backend.
Every test names the reference lines whose behaviour it checks."""
import pickle

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

LOSSES = ("logistic", "warp", "bpr", "warp-kos")


@pytest.fixture(scope="module")
def LightFM():
    from lightfm_amd import LightFM, _native
    assert _native.device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    return LightFM


def _random_interactions(nu, ni, density=0.05, seed=42):
    m = sp.rand(nu, ni, density=density, format="csr", random_state=seed)
    m.data[:] = 1.0
    return m


@pytest.mark.parametrize("loss", LOSSES)
def test_empty_interaction_matrix_fits(LightFM, loss):
    """T_API:10-17."""
    model = LightFM(loss=loss)
    assert model.fit_partial(sp.coo_matrix((10, 100), dtype=np.int32)) is model
    assert model.item_embeddings.shape == (100, 10) and np.isfinite(model.item_embeddings).all()


def test_every_sparse_format_and_dtype_is_coerced(LightFM):
    """T_API:20-54, 95-118: coo / lil / csr / csc x int32 / int64 / float32 / float64."""
    nu, ni, nf = 10, 100, 20
    rng = np.random.RandomState(0)
    for mattype in (sp.coo_matrix, sp.lil_matrix, sp.csr_matrix, sp.csc_matrix):
        for dtype in (np.int32, np.int64, np.float32, np.float64):
            train = mattype((nu, ni), dtype=dtype)
            uf, itf = mattype((nu, nf), dtype=dtype), mattype((ni, nf), dtype=dtype)
            model = LightFM()
            model.fit_partial(train, sample_weight=train.tocoo(), user_features=uf, item_features=itf)
            out = model.predict(rng.randint(0, nu, 10).astype(np.int32), rng.randint(0, ni, 10).astype(np.int32),
                                user_features=uf, item_features=itf)
            assert out.dtype == np.float32 and out.shape == (10,)
            model.predict_rank(train, user_features=uf, item_features=itf)


@pytest.mark.parametrize("loss", ("warp", "bpr", "warp-kos"))
def test_duplicate_coo_entries_are_safe(LightFM, loss):
    """T_API:57-74 (issue 117): the positives CSR is built before the shuffle indices."""
    mat = sp.random(1000, 100, density=0.01, random_state=1).tocoo()
    mat.data[:] = 1
    mat.data = np.concatenate((mat.data, mat.data[:500]))
    mat.row = np.concatenate((mat.row, mat.row[:500]))
    mat.col = np.concatenate((mat.col, mat.col[:500]))
    model = LightFM(loss=loss).fit(mat)
    assert np.isfinite(model.user_embeddings).all()


def test_predict_accepts_scalar_user_and_rejects_garbage(LightFM):
    """T_API:77-92."""
    nu, ni = 10, 100
    model = LightFM().fit_partial(_random_interactions(nu, ni))
    for uid in range(nu):
        a = model.predict(np.repeat(uid, ni), np.arange(ni))
        b = model.predict(uid, np.arange(ni))
        assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        model.predict("foo", np.arange(ni))
    with pytest.raises(ValueError):
        model.predict(np.arange(3), np.arange(4))          # length mismatch, LFM:830-834
    with pytest.raises(ValueError):
        model.predict(np.array([-1]), np.array([0]))        # negative ids, LFM:843-849


def test_feature_matrices_with_too_few_rows_fail(LightFM):
    """T_API:121-133."""
    train = sp.coo_matrix((10, 100), dtype=np.int32)
    with pytest.raises(Exception):
        LightFM().fit_partial(train, user_features=sp.csr_matrix((9, 20), dtype=np.int32),
                              item_features=sp.csr_matrix((99, 20), dtype=np.int32))


def test_ids_beyond_the_fitted_features_fail(LightFM):
    """T_API:136-157: predict through identity features wider than what was fitted."""
    train = sp.coo_matrix((10, 100), dtype=np.int32)
    model = LightFM().fit_partial(train, user_features=sp.csr_matrix((10, 20), dtype=np.int32),
                                  item_features=sp.csr_matrix((100, 20), dtype=np.int32))
    with pytest.raises(ValueError):
        model.predict(np.array([20], dtype=np.int32), np.array([20], dtype=np.int32))


def test_constructor_argument_checks(LightFM):
    """T_API:171-183."""
    for bad in (dict(no_components=-1), dict(user_alpha=-1.0), dict(item_alpha=-1.0)):
        with pytest.raises(AssertionError):
            LightFM(**bad)
    with pytest.raises(ValueError):
        LightFM(max_sampled=-1.0)


def test_sample_weight_validation(LightFM):
    """T_API:186-214."""
    train = sp.coo_matrix(np.array([[0, 1], [0, 1]]))
    model = LightFM()
    with pytest.raises(ValueError):
        model.fit(train, sample_weight=sp.coo_matrix(np.zeros((2, 2))))          # wrong nnz
    with pytest.raises(ValueError):
        model.fit(train, sample_weight=np.zeros(3))                              # not a COO matrix
    with pytest.raises(ValueError):
        model.fit(train, sample_weight=sp.coo_matrix((train.data, (train.row[::-1], train.col[::-1]))))
    model.fit(train, sample_weight=sp.coo_matrix((train.data, (train.row, train.col))))
    with pytest.raises(NotImplementedError):
        LightFM(loss="warp-kos").fit(train, sample_weight=np.ones(1))


def test_predict_rank_known_answers(LightFM):
    """T_API:217-282: ranks are a permutation, train exclusion, pessimistic ties, shape check."""
    nu, ni = 10, 100
    train = _random_interactions(nu, ni, density=0.1)
    model = LightFM().fit_partial(train)
    dense = sp.csr_matrix(np.ones((nu, ni)))
    ranks = model.predict_rank(dense, num_threads=2).toarray()
    for row in range(nu):
        # two items with EXACTLY the same float32 score share a (pessimistic) rank in the reference too: ~2.5 % of the
        # unseeded ten-user models of this test have such a pair (tools/ranks_stress.py, profiles/r04_visit_p.txt)
        if len(np.unique(model.predict(np.repeat(row, ni), np.arange(ni)))) == ni:
            assert np.array_equal(np.sort(ranks[row]), np.arange(ni))
    assert np.all(model.predict_rank(dense, train_interactions=dense, check_intersections=False).toarray() == 0)
    ranks = model.predict_rank(dense, train_interactions=train, check_intersections=False).toarray()
    assert np.array_equal(ranks.max(axis=1), ni - 1 - np.asarray(train.getnnz(axis=1)).ravel())
    with pytest.raises(ValueError):
        model.predict_rank(train, train_interactions=train, check_intersections=True)
    model.predict_rank(train, train_interactions=train, check_intersections=False)
    for name in ("user_embeddings", "item_embeddings", "user_biases", "item_biases"):
        setattr(model, name, np.zeros_like(getattr(model, name)))
    ranks = model.predict_rank(dense, num_threads=2).toarray()
    assert ranks.min() == ni - 1 and ranks.max() == ni - 1
    with pytest.raises(ValueError):
        model.predict_rank(sp.csr_matrix((5, 5)), num_threads=2)


def test_divergence_raises(LightFM):
    """T_API:285-294: the on-device finite check replaces LFM:447-464."""
    with pytest.raises(ValueError):
        LightFM(learning_rate=1e7, loss="warp").fit(_random_interactions(1000, 1000, 0.01), epochs=10)


def test_sklearn_params_roundtrip(LightFM):
    """T_API:297-306."""
    model = LightFM()
    params = model.get_params()
    assert LightFM(**params).get_params() == params
    model.set_params(**params)
    with pytest.raises(ValueError):
        model.set_params(invalid_param=666)


def test_unfitted_model_refuses_to_predict(LightFM):
    """T_API:309-323."""
    model = LightFM()
    with pytest.raises(ValueError):
        model.predict(np.arange(10), np.arange(10))
    with pytest.raises(ValueError):
        model.predict_rank(sp.csr_matrix((3, 3)))
    with pytest.raises(ValueError):
        model.get_user_representations()
    with pytest.raises(ValueError):
        model.get_item_representations()


def test_nan_inputs_raise(LightFM):
    """T_API:326-351."""
    train = _random_interactions(200, 200)
    features = sp.identity(200, format="csr")
    features.data = features.data * np.nan
    with pytest.raises(ValueError):
        LightFM(loss="warp").fit(train, epochs=2, user_features=features, item_features=features)
    bad = train.copy()
    bad.data = bad.data * np.nan
    with pytest.raises(ValueError):
        LightFM(loss="warp").fit(bad)


def test_warp_with_two_items(LightFM):
    """T_API:374-382."""
    model = LightFM(loss="warp", max_sampled=10).fit(_random_interactions(1000, 2, density=0.3))
    assert np.isfinite(model.item_embeddings).all()


@pytest.mark.parametrize("schedule", ("adagrad", "adadelta"))
def test_accumulators_initialise_and_evolve(LightFM, schedule):
    """T_ML:602-652."""
    train = _random_interactions(300, 200)
    model = LightFM(learning_schedule=schedule, loss="warp")
    model.fit_partial(train, epochs=0)
    init = 1.0 if schedule == "adagrad" else 0.0
    assert np.all(model.item_embedding_gradients == init) and np.all(model.user_bias_gradients == init)
    assert np.all(model.item_embedding_momentum == 0)
    model.fit_partial(train, epochs=1)
    assert (model.item_embedding_gradients > init).any() and (model.user_bias_gradients > init).any()
    assert ((model.item_embedding_momentum > 0).any()) == (schedule == "adadelta")


def test_zero_weight_users_accumulate_no_gradient(LightFM):
    """T_ML:437-460, 517-533."""
    train = _random_interactions(200, 150).tocoo()
    weights = train.copy().astype(np.float32)
    zero_users = weights.row < 100
    weights.data[zero_users] = 0.0
    for loss in ("logistic", "warp", "bpr"):
        model = LightFM(loss=loss).fit(train, sample_weight=weights, epochs=2)
        assert np.all(model.user_embedding_gradients[:100] == 1.0)
        assert (model.user_embedding_gradients[100:] > 1.0).any()


def test_fit_resets_and_fit_partial_resumes_and_pickle_roundtrips(LightFM):
    """T_ML:375-412, 463-472."""
    train = _random_interactions(300, 200)
    model = LightFM(loss="warp", random_state=3).fit(train, epochs=2)
    before = model.item_embeddings.copy()
    clone = pickle.loads(pickle.dumps(model))
    assert np.array_equal(clone.item_embeddings, before)
    clone.fit_partial(train, epochs=1)
    assert not np.array_equal(clone.item_embeddings, before)
    model.fit(train, epochs=1)  # fit() starts from scratch (LFM:548)
    assert model.item_embeddings.shape == before.shape


def test_max_sampled_bounds_the_draws(LightFM):
    """doc/examples/warp_loss.rst:157-166 / PYX:857: never more than max_sampled draws per positive."""
    train = _random_interactions(300, 200)
    for ms in (1, 3, 10):
        model = LightFM(loss="warp", max_sampled=ms).fit(train, epochs=2)
        for st in model._last_epoch_stats:
            positives, draws = st["counters"][0], st["counters"][1]
            assert positives == train.nnz and positives <= draws <= ms * positives


def test_representations_reproduce_predictions(LightFM):
    """T_ML:320-351: get_*_representations == what predict uses (within 1e-6)."""
    nu, ni = 60, 40
    train = _random_interactions(nu, ni, density=0.2)
    itf = sp.hstack([sp.identity(ni), sp.random(ni, 7, density=0.3, random_state=2)]).tocsr().astype(np.float32)
    model = LightFM(loss="warp", no_components=12).fit(train, item_features=itf, epochs=3)
    ub, ue = model.get_user_representations()
    ib, ie = model.get_item_representations(itf)
    uids, iids = np.repeat(np.arange(nu), ni).astype(np.int32), np.tile(np.arange(ni), nu).astype(np.int32)
    want = (ue[uids] * ie[iids]).sum(axis=1) + ub[uids] + ib[iids]
    got = model.predict(uids, iids, item_features=itf)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
