"""Host-side logic of lightfm_amd.LightFM that runs before (or without) the device: argument
checks, input coercion, initialisation order, error types -- the reference's contract
(lightfm/lightfm.py, "LFM"; tests/test_api.py, "T_API").  CPU only."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from lightfm_amd import LightFM, _native


def test_constructor_checks():
    """LFM:205-216 / T_API:171-183."""
    for bad in (dict(no_components=0), dict(k=0), dict(n=0), dict(rho=1.0), dict(epsilon=-1.0),
                dict(item_alpha=-0.1), dict(user_alpha=-0.1), dict(learning_schedule="sgd"),
                dict(loss="hinge")):
        with pytest.raises(AssertionError):
            LightFM(**bad)
    with pytest.raises(ValueError):
        LightFM(max_sampled=0)
    m = LightFM(random_state=7)
    assert isinstance(m.random_state, np.random.RandomState)
    rs = np.random.RandomState(3)
    assert LightFM(random_state=rs).random_state is rs


def test_initialisation_order_and_values():
    """LFM:281-312: item table drawn first, then the user table; W = (rand - 0.5) / d as float32,
    biases 0, accumulators 1 (adagrad) or 0 (adadelta), momentum 0."""
    d, ni, nu = 6, 5, 4
    for schedule, g0 in (("adagrad", 1.0), ("adadelta", 0.0)):
        m = LightFM(no_components=d, learning_schedule=schedule, random_state=11)
        m._initialize(d, ni, nu)
        rs = np.random.RandomState(11)
        want_item = ((rs.rand(ni, d) - 0.5) / d).astype(np.float32)
        want_user = ((rs.rand(nu, d) - 0.5) / d).astype(np.float32)
        assert np.array_equal(m.item_embeddings, want_item)
        assert np.array_equal(m.user_embeddings, want_user)
        assert np.all(m.item_embedding_gradients == g0) and np.all(m.user_bias_gradients == g0)
        assert np.all(m.item_biases == 0) and np.all(m.user_embedding_momentum == 0)
        assert all(getattr(m, n).dtype == np.float32 for n in
                   ("item_embeddings", "item_biases", "user_embedding_gradients", "user_bias_momentum"))


def test_feature_matrix_construction():
    """LFM:314-363: identity by default, CSR float32, shape errors."""
    m = LightFM()
    uf, itf = m._construct_feature_matrices(3, 4, None, None)
    assert uf.shape == (3, 3) and itf.shape == (4, 4) and uf.dtype == np.float32 and uf.format == "csr"
    with pytest.raises(Exception):
        m._construct_feature_matrices(3, 4, sp.csr_matrix((2, 5)), None)
    with pytest.raises(Exception):
        m._construct_feature_matrices(3, 4, None, sp.csr_matrix((3, 5)))
    m._initialize(4, 4, 3)
    with pytest.raises(ValueError):  # more feature columns than estimated embeddings
        m._construct_feature_matrices(3, 4, sp.identity(9, format="csr"), None)


def test_positives_lookup_is_sorted_csr():
    """LFM:365-372."""
    coo = sp.coo_matrix((np.ones(5, np.float32), ([0, 0, 1, 0, 1], [4, 1, 3, 2, 0])), shape=(2, 5))
    mat = LightFM()._get_positives_lookup_matrix(coo)
    assert mat.format == "csr" and mat.has_sorted_indices
    assert list(mat.indices[mat.indptr[0]:mat.indptr[1]]) == [1, 2, 4]


def test_sample_weight_processing():
    """LFM:381-420 / T_API:186-214."""
    train = sp.coo_matrix(np.array([[0, 1], [0, 1]], dtype=np.float32))
    m = LightFM()
    assert m._process_sample_weight(train, None) is train.data          # aliases Y when all ones
    two = sp.coo_matrix(np.array([[0, 2], [0, 1]], dtype=np.float32))
    w = m._process_sample_weight(two, None)
    assert w is not two.data and np.all(w == 1.0)
    with pytest.raises(ValueError):
        m._process_sample_weight(train, np.zeros(2))
    with pytest.raises(ValueError):
        m._process_sample_weight(train, sp.coo_matrix(np.zeros((3, 3))))
    with pytest.raises(ValueError):
        m._process_sample_weight(train, sp.coo_matrix((train.data, (train.row[::-1], train.col[::-1]))))
    with pytest.raises(NotImplementedError):
        LightFM(loss="warp-kos")._process_sample_weight(train, sp.coo_matrix(train))
    ok = sp.coo_matrix((np.array([0.5, 2.0]), (train.row, train.col)), shape=train.shape)
    assert m._process_sample_weight(train, ok).dtype == np.float32


def test_errors_raised_before_any_device_work():
    """T_API:309-351, 121-133: not fitted, NaN input, bad thread count, bad shapes."""
    m = LightFM()
    with pytest.raises(ValueError):
        m.predict(np.arange(3), np.arange(3))
    with pytest.raises(ValueError):
        m.get_item_representations()
    train = sp.rand(20, 30, density=0.2, format="coo", random_state=1)
    bad = train.copy()
    bad.data = bad.data * np.nan
    with pytest.raises(ValueError):
        LightFM(loss="warp").fit(bad)
    with pytest.raises(ValueError):
        LightFM().fit(train, num_threads=0)
    with pytest.raises(Exception):
        LightFM().fit(train, item_features=sp.csr_matrix((29, 5)))


def test_sklearn_params():
    """LFM:1049-1107 / T_API:297-306."""
    m = LightFM(no_components=17, loss="bpr", max_sampled=3)
    p = m.get_params()
    assert p["no_components"] == 17 and p["loss"] == "bpr" and p["max_sampled"] == 3
    assert LightFM(**p).get_params() == p
    assert m.set_params(no_components=5) is m and m.no_components == 5
    with pytest.raises(ValueError):
        m.set_params(bogus=1)


def test_fit_without_a_gpu_fails_loudly():
    """No CPU fallback: valid input on a box without a HIP device raises the backend error."""
    if _native.device_count() > 0:
        pytest.skip("a GPU is present")
    train = sp.rand(20, 30, density=0.2, format="coo", random_state=1)
    with pytest.raises(_native.HipBackendError):
        LightFM(loss="warp").fit(train)


def test_merge_schedule_is_global_and_grows_with_history():
    """Every rank derives the same segment list from global numbers; the interval between merges
    grows with the training history from merge_min to merge_max."""
    from lightfm_amd.distributed import MergePolicy, merge_schedule, segment_positions
    pol = MergePolicy(merge_k=4, merge_min=1000, merge_max=50000)
    fr = merge_schedule(0, 200000, 4, pol)
    assert fr[0] == 0.0 and fr[-1] == 1.0 and np.all(np.diff(fr) > 0)
    seg = np.diff(fr) * 200000
    assert abs(seg[0] - 1000) < 1 and seg.max() <= 50000 + 1
    assert np.all(np.diff(seg[:-1]) >= -1)            # never shrinks (the last one is the remainder)
    late = merge_schedule(10_000_000, 200000, 4, pol)  # a trained model: merge_max from the start
    assert len(late) == 5
    # ranks of different shard sizes get the same NUMBER of segments, covering their shard exactly
    for n_local in (49999, 50001, 7):
        pos = segment_positions(fr, n_local)
        assert len(pos) == len(fr) and pos[0] == 0 and pos[-1] == n_local and np.all(np.diff(pos) >= 0)
    assert list(merge_schedule(0, 0, 2, pol)) == [0.0, 1.0]
    auto = merge_schedule(1 << 40, 10 << 20, 8, MergePolicy())  # default merge_max: one launch per rank
    assert len(auto) == 3  # 10 Mi interactions / (8 * 2^20) -> 2 segments


def test_local_shard_rebased_ids():
    from lightfm_amd.distributed import local_shard
    from tests import helpers as H
    coo = H.make_interactions(200, 90, 4000, seed=1)
    full, bounds = local_shard(coo, 1, 3)
    reb, b2 = local_shard(coo, 1, 3, rebase=True)
    assert np.array_equal(bounds, b2)
    assert reb.shape == (bounds[2] - bounds[1], 90) and reb.nnz == full.nnz
    assert np.array_equal(reb.row + bounds[1], full.row) and np.array_equal(reb.col, full.col)


def test_merge_plan_hot_rows_and_full_merges():
    """merge_plan: the full-merge schedule cut into pieces of at most hot_max interactions when the replicated
    side has hot (shared) feature rows; hot_rows: the feature columns an interaction touches with probability >= hot_share."""
    import scipy.sparse as sp
    from lightfm_amd import synthetic
    from lightfm_amd.distributed import MergePolicy, hot_rows, merge_plan, merge_schedule
    pol = MergePolicy()
    n, world = 20_000_000, 8
    full = merge_schedule(3 * n, n, world, pol)
    fr, kinds = merge_plan(3 * n, n, world, pol, 0, has_hot=False)
    assert np.array_equal(fr, full) and set(kinds) == {"full"}
    fr, kinds = merge_plan(3 * n, n, world, pol, 0, has_hot=True)
    assert fr[0] == 0.0 and fr[-1] == 1.0 and np.all(np.diff(fr) > 0) and len(kinds) == len(fr) - 1
    assert np.max(np.diff(fr)) * n <= world * (1 << 17) + 1  # no piece longer than the hot cadence
    full_ends = [f for f, k in zip(fr[1:], kinds) if k == "full"]
    np.testing.assert_allclose(full_ends, full[1:])            # the full merges are where they were
    assert kinds[-1] == "full" and kinds.count("hot") > kinds.count("full")
    # a fresh model: the ramp's short segments are already below the hot cadence
    fr0, kinds0 = merge_plan(0, n, world, pol, 0, has_hot=True)
    assert kinds0[0] == "full" and np.diff(fr0)[0] * n <= 16384 + 1
    # identical on every rank by construction: depends on global numbers only
    assert np.array_equal(merge_plan(3 * n, n, world, pol, 0, True)[0], fr)

    feats = synthetic.tag_item_features(2000, n_tags=50, per_item=4)
    hot = hot_rows(feats, pol.hot_share)
    assert np.array_equal(hot, np.arange(2000, 2050))           # the tag block, ascending; never the identity block
    assert len(hot_rows(None, pol.hot_share)) == 0 and len(hot_rows(sp.identity(30, format="csr"), pol.hot_share)) == 0
    # BASELINE shapes: C3's 1 128 tag rows are hot, C5's hashed feature rows (80 items each of 10 M) are not
    assert len(hot_rows(synthetic.tag_item_features(26744), pol.hot_share)) == 1128
    assert len(hot_rows(synthetic.hashed_item_features(200_000, n_cols=20_000), pol.hot_share)) == 0


def test_weight_array_signature_sees_edits_moves_and_reassignment():
    """LightFM._array_signature guards the device-resident scoring session: a single-cell edit, a swap of two
    rows (same multiset of values), a swap of two columns and a re-assigned array must all change it; an
    untouched array must not."""
    from lightfm_amd import LightFM
    rng = np.random.RandomState(0)
    a = rng.randn(500, 32).astype(np.float32)
    sig = LightFM._array_signature(a)
    assert LightFM._array_signature(a) == sig
    b = a.copy()
    assert LightFM._array_signature(b) != sig            # another object, even with equal content
    assert LightFM._array_signature(b)[3:] == sig[3:]    # ... whose content part agrees
    a[17, 5] = np.nextafter(a[17, 5], np.float32(9))     # one ulp, in place
    assert LightFM._array_signature(a)[3:] != sig[3:]
    a[17, 5] = b[17, 5]
    assert LightFM._array_signature(a) == sig
    a[[3, 4]] = a[[4, 3]]                                # rows swapped in place
    assert LightFM._array_signature(a)[3:] != sig[3:]
    a[[3, 4]] = a[[4, 3]]
    a[:, [0, 1]] = a[:, [1, 0]]                          # columns swapped in place
    assert LightFM._array_signature(a)[3:] != sig[3:]
    v = rng.randn(777).astype(np.float32)                # odd length, one-dimensional (biases)
    sv = LightFM._array_signature(v)
    v[-1] += 1.0
    assert LightFM._array_signature(v)[3:] != sv[3:]
    assert LightFM._array_signature(np.zeros((0, 8), np.float32))[2:] == (0,)


def test_committed_traffic_feeds_the_roofline():
    """bench.py copies roofline.traffic from profiles/traffic.json when the run's dominant kernel is the profiled
    one: the committed file must name the kernels the default configurations actually launch."""
    import importlib.util, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_traffic", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    table = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    # (as bench.kernel_label prints them: <candidates, SHARDED, USTORE, VEC> / <loss id, NC, TIMED, REG, HOT, ADA>)
    names = {"c2": "fit_warp_tile_ahead_kernel<10, false, true, 4>",       # user rows by plain stores (lfm_opts.user_store)
             "c4shard": "fit_warp_tile_ahead_kernel<10, false, false, 4>",
             "c3": "fit_feat_kernel<2, 2, false, false, true, false> + hot_slice_kernel<8>",  # the hot set (plan_flags bit 5)
             "c5shard": "fit_feat_kernel<3, 2, false, false, false, false>"}
    for cfg, kernel in names.items():
        assert os.path.exists(os.path.join(root, table[cfg]["source"]))
        value, source, extra = bench.committed_traffic(cfg, kernel)
        assert value is not None and value > 1e8 and source == table[cfg]["source"], (cfg, value, source)
        # scaled to a run's own launch length: bytes per interaction of the profiled run x interactions per launch
        half, _, extra = bench.committed_traffic(cfg, kernel, table[cfg]["interactions_per_launch_profiled"] / 2)
        assert abs(half / (value / 2) - 1.0) < 1e-9 and 0.1 < extra["traffic_over_algorithmic_profiled"] < 2.5  # (c3: the shared rows live in LDS and L2)
    value, why, _ = bench.committed_traffic("c2", "fit_warp_kernel (generic)")
    assert value is None and "this run's kernel" in why


def test_evaluation_reductions_match_the_sparse_matrix_formulation():
    """lightfm_amd.evaluation reduces the rank CSR's value array with segmented ufunc reductions; the reference
    (lightfm/evaluation.py:14-327) does the same with sparse-matrix operations.  Same values and dtypes, with
    and without preserve_rows, with a user without test interactions -- on a stand-in model (no device)."""
    import scipy.sparse as sp
    from lightfm_amd import evaluation as E
    rng = np.random.RandomState(0)
    test = sp.random(60, 40, density=0.1, format="lil", random_state=1, dtype=np.float32)
    test[7] = 0
    test[59] = 0
    test = test.tocsr()
    test.eliminate_zeros()
    test.data[:] = 1
    rank_values = rng.randint(0, 40, size=test.nnz).astype(np.float32)

    class Model(object):
        def predict_rank(self, t, **kw):
            return sp.csr_matrix((rank_values.copy(), t.indices, t.indptr), shape=t.shape)

    m, has = Model(), test.getnnz(axis=1) > 0

    def precision(k, preserve):
        r = m.predict_rank(test)
        r.data = np.less(r.data, k, r.data)
        p = np.squeeze(np.array(r.sum(axis=1))) / k
        return p if preserve else p[has]

    def recall(k, preserve):
        r = m.predict_rank(test)
        r.data = np.less(r.data, k, r.data)
        hit, relevant = np.squeeze(np.array(r.sum(axis=1))), np.squeeze(test.getnnz(axis=1))
        if not preserve:
            hit, relevant = hit[has], relevant[has]
        with np.errstate(all="ignore"):
            return hit / relevant

    def reciprocal(preserve):
        r = m.predict_rank(test)
        r.data = 1.0 / (r.data + 1.0)
        x = np.squeeze(np.array(r.max(axis=1).todense()))
        return x if preserve else x[has]

    for preserve in (False, True):
        for got, want in ((E.precision_at_k(m, test, k=5, preserve_rows=preserve), precision(5, preserve)),
                          (E.recall_at_k(m, test, k=5, preserve_rows=preserve), recall(5, preserve)),
                          (E.reciprocal_rank(m, test, preserve_rows=preserve), reciprocal(preserve))):
            assert got.dtype == want.dtype and got.shape == want.shape
            assert np.array_equal(got, want, equal_nan=True)
    with pytest.raises(ValueError):
        E.precision_at_k(m, test, num_threads=0)


def test_session_opens_on_the_configured_device(monkeypatch):
    """options.device (LIGHTFM_AMD_DEVICE) is the GPU LightFM's sessions are created on; an explicit
    device argument (DistributedFit, bench.py: the rank's GPU) wins."""
    import lightfm_amd.lightfm as L
    from lightfm_amd._lightfm_fast import CSRMatrix
    from lightfm_amd.options import options
    seen = []

    class FakeLib(object):
        def lfm_session_create(self, handle, device, *rest):
            seen.append(device)
            return 0

        lfm_session_create_scoring = lfm_session_create

        def lfm_session_destroy(self, handle):
            return 0

    monkeypatch.setattr(L.N, "lib", lambda: FakeLib())
    m = L.LightFM(no_components=4)
    m._initialize(4, 3, 2)
    eye3, eye2 = sp.identity(3, dtype=np.float32, format="csr"), sp.identity(2, dtype=np.float32, format="csr")
    options.set(device=5)
    L._Session(m._get_lightfm_data(), CSRMatrix(eye3), CSRMatrix(eye2))
    L._Session(m._get_lightfm_data(), CSRMatrix(eye3), CSRMatrix(eye2), device=2, scoring=True)
    assert seen == [5, 2]


def test_native_table_initialisation_is_numpys_stream():
    """lfm_host_mt19937_table (the embedding initialisation of LFM:281-312 restated natively on the RandomState's own
    state): the same float32 values as ((rand(rows, d) - 0.5) / d).astype(float32) and the same generator state
    afterwards, from fresh, mid-block, odd and end-of-block positions."""
    from lightfm_amd import _native as N
    import os
    if not os.path.exists(N.LIB_PATH):
        pytest.skip("liblfm_hip.so not built")
    for seed, burn, rows, d in ((1, 0, 300, 64), (2, 5, 1000, 17), (3, 311, 2500, 10), (4, 623, 4096, 4), (5, 1248, 20000, 3)):
        a, b = np.random.RandomState(seed), np.random.RandomState(seed)
        if burn:
            a.randint(0, 1 << 30, size=burn), b.randint(0, 1 << 30, size=burn)  # (one 32-bit output each: odd positions)
        got = N.init_table(a, rows, d)
        want = ((b.rand(rows, d) - 0.5) / d).astype(np.float32)
        assert got.dtype == np.float32 and np.array_equal(got, want), (seed, burn)
        sa, sb = a.get_state(), b.get_state()
        assert np.array_equal(sa[1], sb[1]) and sa[2] == sb[2]
        assert a.rand() == b.rand() and a.randint(0, 100) == b.randint(0, 100)
    # the model's initialisation through it: items first, then users, one stream
    from lightfm_amd import LightFM
    m = LightFM(no_components=16, random_state=9)
    m._initialize(16, 1500, 2500)
    r = np.random.RandomState(9)
    assert np.array_equal(m.item_embeddings, ((r.rand(1500, 16) - 0.5) / 16).astype(np.float32))
    assert np.array_equal(m.user_embeddings, ((r.rand(2500, 16) - 0.5) / 16).astype(np.float32))


def test_native_input_scan_matches_numpy():
    from lightfm_amd import _native as N
    import os
    if not os.path.exists(N.LIB_PATH):
        pytest.skip("liblfm_hip.so not built")
    a = np.ones(3_000_000, np.float32)
    assert N.host_scan(a) == (True, True)
    a[12345] = 0.5
    assert N.host_scan(a) == (False, True)
    for bad in (np.inf, -np.inf, np.nan):
        a[777] = bad
        assert N.host_scan(a) == (False, False)
    a[:] = 3e38   # finite values whose float32 sum is not (the reference tests isfinite(sum))
    assert N.host_scan(a) == (False, False)


@pytest.mark.parametrize("preset", [None, "0", "1"])
def test_distributed_fit_asks_for_the_rank_independent_row_stride(monkeypatch, preset):
    """csrc/session.hip pads a training session's rows by what ITS feature matrices look like (LIGHTFM_AMD_ROW_ALIGN = 1);
    the ranks of a job exchange item rows, so DistributedFit creates its session under mode 2 (the width-only rule) --
    unless padding is switched off -- and leaves the variable as it found it, also when the session cannot be created."""
    import scipy.sparse as sp
    import lightfm_amd.lightfm as L
    from lightfm_amd import LightFM
    from lightfm_amd.distributed import DistributedFit
    seen = []

    class Stop(Exception):
        pass

    class FakeSession(object):
        def __init__(self, *args, **kw):
            seen.append(os.environ.get("LIGHTFM_AMD_ROW_ALIGN"))
            raise Stop()

    monkeypatch.setattr(L, "_Session", FakeSession)
    if preset is None:
        monkeypatch.delenv("LIGHTFM_AMD_ROW_ALIGN", raising=False)
    else:
        monkeypatch.setenv("LIGHTFM_AMD_ROW_ALIGN", preset)
    rng = np.random.RandomState(0)
    coo = sp.coo_matrix((np.ones(200, np.float32), (rng.randint(0, 50, 200), rng.randint(0, 30, 200))), shape=(50, 30))
    with pytest.raises(Stop):
        DistributedFit(LightFM(no_components=20, loss="warp", random_state=1), coo, 0, 1)
    assert seen == ["0" if preset == "0" else "2"]
    assert os.environ.get("LIGHTFM_AMD_ROW_ALIGN") == preset
