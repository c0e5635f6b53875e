"""GPU regression tests of the round-3 fixes for the process abort (DESIGN.md "Root cause of the round-2
process abort"): the device-memory pool (csrc/pool.hpp), the in-kernel shuffle guard (device.hpp: guard_row ->
LFM_ECORRUPT instead of a GPU memory fault that aborts the interpreter) and the id range checks at upload."""
import numpy as np
import pytest
import scipy.sparse as sp

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reset_options():
    from lightfm_amd.options import options
    defaults = dict(mode="parallel", launches_per_epoch=0, first_batch=0, max_waves=0, log_samples=False,
                    warp_kernel=0, feat_kernel=0, ramp_k=0, update_mode=0, shared_cap=0, host_positives=False, debug=0)
    options.set(**defaults)
    yield
    options.set(**defaults)


def _open(model, coo, n_users, n_items):
    from lightfm_amd._lightfm_fast import CSRMatrix
    from lightfm_amd.lightfm import _Session
    s = _Session(model._get_lightfm_data(), CSRMatrix(H.identity_features(n_items)), CSRMatrix(H.identity_features(n_users)))
    s.set_interactions(None, np.ascontiguousarray(coo.row), np.ascontiguousarray(coo.col), coo.data, coo.data)
    s.build_positives(n_users, n_items)
    return s


def test_buffers_of_a_closed_session_are_reused_not_returned_to_the_runtime():
    from lightfm_amd import LightFM, _native
    coo = H.make_interactions(700, 500, 20000, seed=11)
    m = LightFM(no_components=32, loss="warp", random_state=2)
    m.fit(coo, epochs=1)  # warms the pool with this shape's size classes
    reserved0, _ = _native.device_pool_stats()
    for _ in range(5):
        LightFM(no_components=32, loss="warp", random_state=2).fit(coo, epochs=1)
    reserved1, cached1 = _native.device_pool_stats()
    assert reserved1 == reserved0, "a second model of the same shape must not take new memory from the runtime"
    assert cached1 > 0
    released = _native.device_trim()
    assert released == cached1
    assert _native.device_pool_stats() == (reserved1 - released, 0)
    LightFM(no_components=32, loss="warp", random_state=2).fit(coo, epochs=1)  # works again after a trim


@pytest.mark.parametrize("family", ["tile", "generic", "row-stream", "serial", "bpr-tile"])
def test_shuffle_entry_out_of_range_is_an_error_not_a_gpu_fault(family, monkeypatch):
    """A shuffle slot that is not a permutation of [0, n) used to index the COO out of range (GPU memory fault,
    SIGABRT of the process).  Every kernel family now clamps the index and the epoch fails with LFM_ECORRUPT."""
    from lightfm_amd import LightFM, options
    from lightfm_amd._lightfm_fast import make_opts
    from lightfm_amd._native import HipBackendError
    nu, ni = 400, 300
    coo = H.make_interactions(nu, ni, 9000, seed=6)
    loss = "bpr" if family in ("row-stream", "bpr-tile") else "warp"
    if family == "row-stream":  # (identity BPR runs the tile kernel's BPR instantiation otherwise: the fifth family)
        monkeypatch.setenv("LIGHTFM_AMD_BPR_WIDE_TILE", "0")
    if family == "generic":
        options.set(warp_kernel=1, feat_kernel=1)
    if family == "serial":
        options.set(mode="serial")
    m = LightFM(no_components=32, loss=loss, random_state=1)
    m._initialize(32, ni, nu)
    s = _open(m, coo, nu, ni)
    try:
        n = coo.nnz
        shuffle = np.arange(n, dtype=np.int32)
        shuffle[n // 3] = 0x3f800000  # what the round-2 abort had read there
        shuffle[n // 2] = -7
        s.upload_shuffle(shuffle)
        seeds = np.array([5], np.uint32)
        o, _ = make_opts()
        with pytest.raises(HipBackendError, match="shuffle entry outside"):
            s.epoch(loss, 0.0, 0.0, 5, 10, seeds, o)
        want = {"tile": 1, "generic": 0, "row-stream": 2, "serial": 0, "bpr-tile": 1}[family]
        assert o.kernel_used == want
        # the session stays usable: a valid slot trains
        s.upload_shuffle(np.arange(n, dtype=np.int32))
        o, _ = make_opts()
        s.epoch(loss, 0.0, 0.0, 5, 10, seeds, o)
        assert o.counters[0] == n
    finally:
        s.close()


def test_ids_out_of_range_are_rejected_at_upload():
    from lightfm_amd import LightFM
    nu, ni = 300, 200
    coo = H.make_interactions(nu, ni, 4000, seed=3)
    m = LightFM(no_components=16, loss="warp", random_state=1)
    m._initialize(16, ni, nu)
    from lightfm_amd._lightfm_fast import CSRMatrix
    from lightfm_amd.lightfm import _Session
    s = _Session(m._get_lightfm_data(), CSRMatrix(H.identity_features(ni)), CSRMatrix(H.identity_features(nu)))
    try:
        rows, cols = coo.row.copy(), coo.col.copy()
        cols[17] = ni  # one past the last item
        with pytest.raises(ValueError, match="item_ids"):
            s.set_interactions(None, rows, cols, coo.data, coo.data)
        rows[5] = -1
        with pytest.raises(ValueError, match="user_ids"):
            s.set_interactions(None, rows, coo.col.copy(), coo.data, coo.data)
    finally:
        s.close()
    # a feature matrix whose column ids exceed the embedding table
    feats = sp.csr_matrix((np.ones(ni, np.float32), (np.arange(ni), np.arange(ni))), shape=(ni, ni)).astype(np.float32)
    feats.indices = feats.indices.copy()
    feats.indices[3] = ni + 40
    feats.data[0] = 2.0  # not an identity matrix: its indices are really read
    with pytest.raises(ValueError, match="item_features.indices"):
        _Session(m._get_lightfm_data(), CSRMatrix(feats), CSRMatrix(H.identity_features(nu)))


def test_uncached_tables_then_alpha_one_launch_storm_in_one_process():
    """The sequence at which every unpatched run of the suite died: models with large (uncached) tables are
    fitted and freed, then an alpha = 1 WARP model is fitted through the generic kernel."""
    from lightfm_amd import LightFM
    rng = np.random.RandomState(5)
    big = sp.coo_matrix((np.ones(200000, np.float32), (rng.randint(0, 9000, 200000).astype(np.int32),
                                                       rng.randint(0, 3000, 200000).astype(np.int32))), shape=(9000, 3000))
    big.sum_duplicates()
    big = sp.coo_matrix((np.ones(big.nnz, np.float32), (big.row, big.col)), shape=big.shape)
    small = H.make_interactions(1500, 900, 60000, seed=2)
    for rep in range(3):
        LightFM(no_components=64, loss="warp", learning_schedule="adadelta", random_state=3).fit(big, epochs=1)
        m = LightFM(no_components=10, loss="warp", item_alpha=1.0, user_alpha=1.0, random_state=10)
        m.fit_partial(small, epochs=2)
        assert np.isfinite(m.item_embeddings).all() and np.isfinite(m.user_embeddings).all()
