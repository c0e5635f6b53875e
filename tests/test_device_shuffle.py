"""The keyed on-device epoch shuffle (lfm_session_device_shuffle): a bijection of [0, n), the same
on host and device, different per key, without obvious structure."""
import ctypes as C

import numpy as np
import pytest


def _perm(n, k0, k1):
    from lightfm_amd import _native as N
    out = np.empty(n, np.int32)
    N.check(N.lib().lfm_shuffle_permutation(N.i32p(out), C.c_int64(n), C.c_uint32(k0), C.c_uint32(k1)))
    return out


@pytest.mark.parametrize("n", [0, 1, 2, 3, 17, 256, 1000, 65536, 65537, 1000003])
def test_permutation_is_a_bijection(n):
    p = _perm(n, 11, 22)
    assert np.array_equal(np.sort(p), np.arange(n, dtype=np.int32))


def test_permutation_depends_on_both_keys_and_looks_random():
    n = 200000
    a, b, c = _perm(n, 1, 2), _perm(n, 1, 3), _perm(n, 4, 2)
    assert (a != b).mean() > 0.99 and (a != c).mean() > 0.99
    x = np.arange(n, dtype=np.float64)
    assert abs(np.corrcoef(x, a)[0, 1]) < 0.01            # no trend
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 0.01    # neighbours unrelated
    step = np.abs(np.diff(a.astype(np.int64))).mean()
    assert abs(step / n - 1.0 / 3.0) < 0.01               # E|U1 - U2| = n/3 for a uniform shuffle
    # every tenth of the output draws evenly from every tenth of the input
    table = np.histogram2d(np.arange(n) * 10 // n, a.astype(np.int64) * 10 // n, bins=10)[0]
    assert np.abs(table / (n / 100.0) - 1.0).max() < 0.1


@pytest.mark.gpu
def test_device_slot_equals_host_restatement():
    import scipy.sparse as sp
    from lightfm_amd import LightFM, _native
    from lightfm_amd._lightfm_fast import CSRMatrix
    from lightfm_amd.lightfm import _Session
    from tests import helpers as H
    assert _native.device_count() > 0
    coo = H.make_interactions(300, 200, 5000, seed=2)
    m = LightFM(no_components=8, loss="warp", random_state=1)
    m._initialize(8, 200, 300)
    s = _Session(m._get_lightfm_data(), CSRMatrix(H.identity_features(200)), CSRMatrix(H.identity_features(300)))
    try:
        s.set_interactions(CSRMatrix(H.positives_csr(coo)), np.ascontiguousarray(coo.row),
                           np.ascontiguousarray(coo.col), coo.data, coo.data)
        s.device_shuffle(123, 456)
        got = s.download_shuffle(coo.nnz)
    finally:
        s.close()
    assert np.array_equal(got, _perm(coo.nnz, 123, 456))


@pytest.mark.gpu
def test_fit_advances_random_state_with_device_shuffle():
    """The reference's tests/test_movielens.py:669-682: the caller's RandomState moves every epoch."""
    from lightfm_amd import LightFM
    from tests import helpers as H
    coo = H.make_interactions(100, 80, 1500, seed=4)
    model = LightFM(learning_rate=0.05, loss="warp", random_state=10)
    model.fit_partial(coo, epochs=1)
    state = model.random_state.get_state()[1].copy()
    model.fit_partial(coo, epochs=1)
    assert not np.all(state == model.random_state.get_state()[1])
    assert np.isfinite(model.item_embeddings).all()
