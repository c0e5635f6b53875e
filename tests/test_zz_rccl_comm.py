"""RCCL plumbing on one GPU: a communicator of ONE rank exercises librccl loading,
ncclCommInitRank, the grouped all-reduces of the three merge modes, the flag reduction and the barrier (the N > 1
arithmetic itself is covered on CPU by tests/test_multi_gpu_semantics.py).

The file is named to be collected LAST: librccl stays loaded (with its runtime threads) for the life of the
process that initialised it; a rank process of a real multi-GPU job carries it from the start, the single-GPU
suite should not run under it."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_roundtrip():
    from lightfm_amd import LightFM, _native as N
    from lightfm_amd._lightfm_fast import CSRMatrix, make_opts
    from lightfm_amd.lightfm import _Session
    from tests import helpers as H
    assert N.device_count() > 0
    coo = H.make_interactions(300, 200, 6000, seed=8)
    m = LightFM(no_components=32, loss="warp", random_state=3)
    m._initialize(32, 200, 300)
    struct = m._get_lightfm_data()
    s = _Session(struct, CSRMatrix(H.identity_features(200)), CSRMatrix(H.identity_features(300)))
    try:
        s.set_interactions(CSRMatrix(H.positives_csr(coo)), np.ascontiguousarray(coo.row),
                           np.ascontiguousarray(coo.col), coo.data, coo.data)
        uid = C.create_string_buffer(N.UNIQUE_ID_BYTES)
        N.check(N.lib().lfm_comm_unique_id(uid))
        s.comm_init(uid, 0, 1)
        before = m.user_embeddings.copy()
        for e in range(2):
            s.device_shuffle(5 + e, 7)
            opts, _ = make_opts()
            s.epoch("warp", 0.0, 0.0, 5, 10, np.array([11 + e], np.uint32), opts)
            assert opts.counters[0] == coo.nnz
        s.comm_barrier()
        s.sync_to_host(struct)
        trained = m.user_embeddings.copy()
        items = m.item_embeddings.copy()
        for mode in (N.MERGE_SUM, N.MERGE_MEAN, N.MERGE_ADAGRAD):
            s.comm_merge(1, mode)   # X := X_start + allreduce(X - X_start) over ONE rank: the identity
            s.sync_to_host(struct)
            np.testing.assert_allclose(m.item_embeddings, items, rtol=1e-6, atol=1e-7)
            items = m.item_embeddings.copy()
        # the sparse merge through RCCL (byte-map all-reduce MAX, packed all-reduces on the communication stream):
        # over ONE rank it is the identity too, synchronous or overlapped
        for e, (mode, overlap) in enumerate([(N.MERGE_SUM, False), (N.MERGE_ADAGRAD, False), (N.MERGE_ADAGRAD, True),
                                             (N.MERGE_MEAN, True)]):
            s.device_shuffle(50 + e, 9)
            opts, _ = make_opts()
            opts.history = 1 << 30
            s.epoch("warp", 0.0, 0.0, 5, 10, np.array([40 + e], np.uint32), opts)
            s.sync_to_host(struct)
            items, acc = m.item_embeddings.copy(), m.item_embedding_gradients.copy()
            trained = m.user_embeddings.copy()
            nbytes = s.comm_merge_sparse(1, mode, overlap)
            assert nbytes >= 200  # the byte map at least
            s.comm_merge_flush()
            s.sync_to_host(struct)
            np.testing.assert_allclose(m.item_embeddings, items, rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(m.item_embedding_gradients, acc, rtol=1e-6, atol=1e-7)
        assert s.comm_any(False) is False and s.comm_any(True) is True
    finally:
        s.close()
    assert not np.array_equal(before, trained)
    assert np.array_equal(m.user_embeddings, trained)  # identity user features: never communicated
    assert np.isfinite(m.item_embeddings).all()
