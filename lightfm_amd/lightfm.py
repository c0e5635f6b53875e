"""`LightFM` estimator with the reference's public API (lightfm/lightfm.py, "LFM"
below) whose epochs run on the MI355X.

Everything a caller can observe is kept: constructor arguments and assertions
(LFM:189-241), the 12 weight attributes and their initialisation order
(LFM:245-312), input coercion and the exceptions raised for bad input
(LFM:314-420, 617-652, 819-849), pickling, `fit_partial` resuming, sklearn
`get_params/set_params`.

`random_state`: when the caller hands in a `numpy.random.RandomState` INSTANCE (a stream it
may share with other code), every epoch consumes it exactly like the reference -- one
`shuffle(arange(n))` then one `randint(size=num_threads)` (LFM:689-690 +
_lightfm_fast.pyx.template:812-814) -- so the stream is in the same state after `fit`.  When
the model owns its stream (seed or None) the epoch's visiting order is instead built on the
device from one `randint(size=624)` block (`options.device_shuffle`, a keyed permutation:
lfm_session_device_shuffle), because numpy's sequential Fisher-Yates over 20 M entries costs
ten device epochs; `options.device_shuffle = False` restores the reference's draws there too.

What changes is below the boundary: instead of handing host arrays to the Cython
extension once per epoch, `fit_partial` opens ONE device-resident session
(include/lfm_hip.h: lfm_session_*), uploads weights / feature CSRs / the COO once, builds the
positives lookup on the device (lfm_session_build_positives instead of LFM:365-372's host
`tocsr()`), runs every epoch on the GPU (only the keys or the shuffled index list and the seeds
travel per epoch), checks finiteness on device and copies the weights back at the end.
"""
import contextlib
import ctypes as C
import threading

import numpy as np
import scipy.sparse as sp

from . import _native as N
from ._lightfm_fast import CSRMatrix, FastLightFM, make_opts, predict_lightfm, predict_ranks
from .options import options

__all__ = ["LightFM"]

CYTHON_DTYPE = np.float32

_WEIGHTS = ("item_embeddings", "item_embedding_gradients", "item_embedding_momentum",
            "item_biases", "item_bias_gradients", "item_bias_momentum",
            "user_embeddings", "user_embedding_gradients", "user_embedding_momentum",
            "user_biases", "user_bias_gradients", "user_bias_momentum")

_NOT_FINITE = ("Not all estimated parameters are finite, your model may have diverged. "
               "Try decreasing the learning rate or normalising feature values and sample weights")


class _Stages(object):
    """LIGHTFM_AMD_TIMING=1: wall time of the stages of fit_partial on stderr (tools/fit_timing.py)."""

    def __init__(self):
        import os
        self.on = os.environ.get("LIGHTFM_AMD_TIMING", "0") not in ("", "0")
        self.marks = []
        if self.on:
            import time
            self.clock = time.perf_counter
            self.t = self.clock()

    def mark(self, name):
        if self.on:
            now = self.clock()
            self.marks.append((name, now - self.t))
            self.t = now

    def report(self):
        if self.on:
            import sys
            print("[lightfm_amd timing] " + ", ".join("%s %.1f ms" % (n, 1e3 * dt) for n, dt in self.marks),
                  file=sys.stderr, flush=True)


def _printed_epochs(n):
    for epoch in range(n):
        print("Epoch {}".format(epoch))
        yield epoch


def _pair_ids(user_ids, item_ids):
    """predict()'s id arguments as two int32 arrays of one length.  Accepted like the reference (LFM:821-840): one
    int user id for many items, lists / tuples, integer arrays of any width."""
    if isinstance(user_ids, int):
        user_ids = np.full(len(item_ids), user_ids, dtype=np.int32)
    pair = [np.asarray(v, dtype=np.int32) if isinstance(v, (list, tuple)) else v for v in (user_ids, item_ids)]
    if len(pair[0]) != len(pair[1]):
        raise ValueError("Expected the number of user IDs (%d) to equal the number of item IDs (%d)"
                         % (len(pair[0]), len(pair[1])))
    return [v if v.dtype == np.int32 else v.astype(np.int32) for v in pair]


class _Session(object):
    """RAII wrapper of lfm_session (include/lfm_hip.h)."""

    def __init__(self, model_struct, item_features, user_features, device=None, scoring=False):
        """scoring=True: only embeddings and biases are uploaded (lfm_session_create_scoring); such a
        session serves predict / predict_ranks / representations and cannot train.  device=None: the
        GPU named by options.device (LIGHTFM_AMD_DEVICE)."""
        if device is None:
            device = int(options.device)
        self.handle = C.c_void_p()
        self._keep = (model_struct, item_features, user_features)
        create = N.lib().lfm_session_create_scoring if scoring else N.lib().lfm_session_create
        N.check(create(C.byref(self.handle), device, model_struct.byref(), item_features.byref(),
                       user_features.byref()))

    def set_features(self, item_features, user_features):
        """Replaces the resident feature matrices (None keeps a side's)."""
        N.check(N.lib().lfm_session_set_features(
            self.handle, item_features.byref() if item_features is not None else None,
            user_features.byref() if user_features is not None else None))
        self._keep = (self._keep[0], self._keep[1] if item_features is None else item_features,
                      self._keep[2] if user_features is None else user_features)

    def predict(self, user_ids, item_ids, predictions):
        N.check(N.lib().lfm_session_predict(
            self.handle, N.i32p(N.require(user_ids, np.int32, 1, "user_ids")),
            N.i32p(N.require(item_ids, np.int32, 1, "item_ids")),
            N.f32p(N.require(predictions, np.float32, 1, "predictions")), C.c_int64(len(predictions))))

    def predict_ranks(self, test, train, ranks):
        N.check(N.lib().lfm_session_predict_ranks(self.handle, test.byref(), train.byref(),
                                                  N.f32p(N.require(ranks, np.float32, 1, "ranks"))))

    def set_interactions(self, positives, rows, cols, data, weight):
        n = len(rows)
        # raw pointers go to the library: hold the typed-memoryview contract here
        rows = N.require(rows, np.int32, 1, "user_ids")
        if cols is not None:
            cols = N.require(cols, np.int32, 1, "item_ids")
        if data is not None:
            data = N.require(data, np.float32, 1, "Y")
        if weight is not None:
            weight = N.require(weight, np.float32, 1, "sample_weight")
        N.check(N.lib().lfm_session_set_interactions(
            self.handle, positives.byref() if positives is not None else None, N.i32p(rows),
            N.i32p(cols), N.f32p(data), N.f32p(weight), C.c_int64(n)))

    def upload_shuffle(self, shuffle, slot=0):
        N.check(N.lib().lfm_session_upload_shuffle(self.handle, slot, N.i32p(shuffle),
                                                   C.c_int64(len(shuffle))))

    def device_shuffle(self, key0, key1, slot=0):
        N.check(N.lib().lfm_session_device_shuffle(self.handle, slot, C.c_uint32(key0), C.c_uint32(key1)))

    def device_shuffle_ahead(self, key0, key1, slot):
        """The permutation of a LATER epoch written while the current one trains (lfm_session_device_shuffle_ahead)."""
        N.check(N.lib().lfm_session_device_shuffle_ahead(self.handle, slot, C.c_uint32(key0), C.c_uint32(key1)))

    def download_shuffle(self, n, slot=0):
        out = np.empty(n, np.int32)
        N.check(N.lib().lfm_session_download_shuffle(self.handle, slot, N.i32p(out), C.c_int64(n)))
        return out

    def epoch(self, loss, item_alpha, user_alpha, k, n, seeds, opts, slot=0):
        N.check(N.lib().lfm_session_epoch(
            self.handle, N.LOSS_IDS[loss], slot, C.c_double(item_alpha), C.c_double(user_alpha),
            C.c_int32(k), C.c_int32(n), N.u32p(seeds), 0 if seeds is None else len(seeds),
            C.byref(opts)))

    def check_finite(self):
        return bool(N.check(N.lib().lfm_session_check_finite(self.handle)))

    def sync_to_host(self, model_struct):
        N.check(N.lib().lfm_session_sync_to_host(self.handle, model_struct.byref()))

    def comm_init(self, unique_id, rank, nranks):
        N.check(N.lib().lfm_session_comm_init(self.handle, unique_id, rank, nranks))

    def comm_merge(self, sides=1, mode=0):
        N.check(N.lib().lfm_session_comm_merge(self.handle, sides, mode))

    def comm_merge_sparse(self, sides=1, mode=0, overlap=True):
        """The merge over the rows touched since the last one (lfm_session_comm_merge_sparse); returns the
        bytes this rank handed to RCCL.  With overlap the result lands at the next merge or flush."""
        nbytes = C.c_int64()
        N.check(N.lib().lfm_session_comm_merge_sparse(self.handle, sides, mode, int(bool(overlap)), C.byref(nbytes)))
        return nbytes.value

    def set_merge_dense_fraction(self, fraction):
        """From which share of a side's rows in a merge's union the next sparse merges carry every row without detecting
        them (default 0.9; > 1 never, <= 0 always)."""
        N.check(N.lib().lfm_session_set_merge_dense_fraction(self.handle, C.c_float(fraction)))

    def comm_merge_flush(self):
        N.check(N.lib().lfm_session_comm_merge_flush(self.handle))

    def set_hot_rows(self, side, rows):
        """The feature rows of `side` merged at the short cadence (ascending int32; lfm_session_set_hot_rows)."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        N.check(N.lib().lfm_session_set_hot_rows(self.handle, side, N.i32p(rows), C.c_int64(len(rows))))

    def comm_merge_hot(self, sides=1, mode=0, overlap=False):
        nbytes = C.c_int64()
        N.check(N.lib().lfm_session_comm_merge_hot(self.handle, sides, mode, int(bool(overlap)), C.byref(nbytes)))
        return nbytes.value

    def merge_begin(self, sides=1):
        N.check(N.lib().lfm_session_merge_begin(self.handle, sides))

    @staticmethod
    def merge_local(sessions, sides=1, mode=0):
        """K sessions of this process on one device merged like K ranks (no RCCL)."""
        arr = (C.c_void_p * len(sessions))(*[s.handle for s in sessions])
        N.check(N.lib().lfm_sessions_merge_local(arr, len(sessions), sides, mode))

    @staticmethod
    def merge_local_sparse(sessions, sides=1, mode=0, overlap=False):
        arr = (C.c_void_p * len(sessions))(*[s.handle for s in sessions])
        N.check(N.lib().lfm_sessions_merge_local_sparse(arr, len(sessions), sides, mode, int(bool(overlap))))

    @staticmethod
    def merge_local_hot(sessions, sides=1, mode=0, overlap=False):
        arr = (C.c_void_p * len(sessions))(*[s.handle for s in sessions])
        N.check(N.lib().lfm_sessions_merge_local_hot(arr, len(sessions), sides, mode, int(bool(overlap))))

    @staticmethod
    def share_items_local(sessions):
        """Owner-sharded item tables over K sessions of one device (lfm_sessions_share_items_local): session j owns
        the item rows [j * rps, (j + 1) * rps), rps = ceil(n_items / K); every session trains against the owners' rows."""
        arr = (C.c_void_p * len(sessions))(*[s.handle for s in sessions])
        N.check(N.lib().lfm_sessions_share_items_local(arr, len(sessions)))

    def export_items(self):
        """The HIP IPC handles of this session's item-side allocations as bytes (lfm_session_export_items)."""
        ex = N.LfmItemExport()
        N.check(N.lib().lfm_session_export_items(self.handle, C.byref(ex)))
        return bytes(bytearray(ex))

    def share_items_ipc(self, exports, my_rank):
        """Owner-sharded item tables over K PROCESSES (lfm_session_share_items_ipc): `exports` = every rank's
        export_items() bytes in rank order, exports[my_rank] this process's own."""
        arr = (N.LfmItemExport * len(exports))()
        for j, blob in enumerate(exports):
            if len(blob) != C.sizeof(N.LfmItemExport):
                raise ValueError("export %d has %d bytes, expected %d" % (j, len(blob), C.sizeof(N.LfmItemExport)))
            C.memmove(C.byref(arr[j]), blob, len(blob))
        N.check(N.lib().lfm_session_share_items_ipc(self.handle, arr, len(exports), my_rank))

    def gather_shared_items(self):
        N.check(N.lib().lfm_session_gather_shared_items(self.handle))

    @staticmethod
    def merge_local_flush(sessions):
        arr = (C.c_void_p * len(sessions))(*[s.handle for s in sessions])
        N.check(N.lib().lfm_sessions_merge_local_flush(arr, len(sessions)))

    def comm_any(self, flag):
        return bool(N.check(N.lib().lfm_session_comm_any(self.handle, int(bool(flag)))))

    def comm_barrier(self):
        N.check(N.lib().lfm_session_comm_barrier(self.handle))

    def load_model(self, model_struct):
        N.check(N.lib().lfm_session_load_model(self.handle, model_struct.byref()))

    def build_positives(self, n_users, n_items):
        N.check(N.lib().lfm_session_build_positives(self.handle, n_users, n_items))

    def download_positives(self, n_users):
        nnz = C.c_int64()
        N.check(N.lib().lfm_session_download_positives(self.handle, None, None, C.byref(nnz)))
        indptr = np.empty(n_users + 1, np.int32)
        indices = np.empty(nnz.value, np.int32)
        N.check(N.lib().lfm_session_download_positives(self.handle, N.i32p(indptr), N.i32p(indices),
                                                       C.byref(nnz)))
        return indptr, indices

    def representations(self, side, features):
        """(biases, embeddings) of every row of `features` (CSRMatrix), side 0 item / 1 user."""
        d = self._keep[0].no_components
        biases = np.empty(features.rows, np.float32)
        emb = np.empty((features.rows, d), np.float32)
        N.check(N.lib().lfm_session_representations(self.handle, side, features.byref(), N.f32p(biases),
                                                    N.f32p(emb)))
        return biases, emb

    def close(self):
        if self.handle:
            N.lib().lfm_session_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LightFM(object):
    """Hybrid latent-representation recommender (API of LFM:24-1107).

    Parameters are those of the reference: no_components, k, n, learning_schedule
    ('adagrad' | 'adadelta'), loss ('logistic' | 'bpr' | 'warp' | 'warp-kos'),
    learning_rate, rho, epsilon, item_alpha, user_alpha, max_sampled, random_state.
    """

    def __init__(self, no_components=10, k=5, n=10, learning_schedule="adagrad", loss="logistic",
                 learning_rate=0.05, rho=0.95, epsilon=1e-6, item_alpha=0.0, user_alpha=0.0,
                 max_sampled=10, random_state=None):
        # LFM:205-216
        assert item_alpha >= 0.0
        assert user_alpha >= 0.0
        assert no_components > 0
        assert k > 0
        assert n > 0
        assert 0 < rho < 1
        assert epsilon >= 0
        assert learning_schedule in ("adagrad", "adadelta")
        assert loss in ("logistic", "warp", "bpr", "warp-kos")
        if max_sampled < 1:
            raise ValueError("max_sampled must be a positive integer")

        self.loss = loss
        self.learning_schedule = learning_schedule
        self.no_components = no_components
        self.learning_rate = learning_rate
        self.k = int(k)
        self.n = int(n)
        self.rho = rho
        self.epsilon = epsilon
        self.max_sampled = max_sampled
        self.item_alpha = item_alpha
        self.user_alpha = user_alpha

        # a RandomState INSTANCE may be shared with the caller's other code: consume it exactly
        # like the reference (see the module docstring)
        self._shared_random_state = isinstance(random_state, np.random.RandomState)
        if random_state is None:
            self.random_state = np.random.RandomState()
        elif isinstance(random_state, np.random.RandomState):
            self.random_state = random_state
        else:
            self.random_state = np.random.RandomState(random_state)

        self._reset_state()

    # ------------------------------------------------------------------ state

    def _reset_state(self):
        self._trained_interactions = 0  # drives the concurrency ramp (lfm_opts.history)
        self._drop_scoring_session()
        for name in _WEIGHTS:
            setattr(self, name, None)

    # ------------------------------------------------- resident scoring session
    # predict / predict_rank of the reference hand the whole FastLightFM to every native call
    # (LFM:862-870, 979-987).  Here the embeddings and biases stay on the device between calls: a
    # scoring session is kept on the model and reused while the four arrays it was built from are the
    # same objects with the same content (a checksum per call: ~4 ms for the ML-20M tables, against
    # re-uploading 43 MB -- or 172 MB with the accumulators, as round 2 did).

    _SCORED = ("item_embeddings", "item_biases", "user_embeddings", "user_biases")

    def _scoring_lock(self):
        lock = self.__dict__.get("_scoring_mutex")
        if lock is None:
            lock = self.__dict__.setdefault("_scoring_mutex", threading.Lock())
        return lock

    def _drop_scoring_session(self):
        if self.__dict__.get("_scoring_owner") == threading.get_ident():
            # called from INSIDE this thread's own `with _scoring_session(...)` block (e.g. fit_partial from a scoring
            # callback): the session in use is closed when the block ends, not under it -- and nobody deadlocks
            self.__dict__["_scoring_drop_pending"] = True
            return
        with self._scoring_lock():  # never under a predict call that is using it
            cached = self.__dict__.pop("_scoring", None)
            if cached is not None:
                cached[0].close()

    @staticmethod
    def _array_signature(a):
        """Identity + content of a weight array: where it lives, its shape, and sum of word[i] * (2 i + 1) modulo 2^64 over
        its bit patterns -- exact (any single-cell edit changes it) and position-sensitive (row / column moves change it).
        Computed natively by a few threads (N.checksum: 1-2 ms for the ML-20M tables); no BLAS call: a matrix-vector
        probe woke every core of a 256-core box per predict call, and under a container CPU quota that throttled the process."""
        if a.size == 0:
            return (id(a), a.shape, 0)
        if a.flags.c_contiguous and a.dtype == np.float32:
            exact = N.checksum(a)
        else:
            exact = hash(a.tobytes())
        return (id(a), a.__array_interface__["data"][0], a.shape, exact)

    @contextlib.contextmanager
    def _scoring_session(self, item_features, user_features):
        """A device session holding the current embeddings and biases, with the given feature matrices.

        The cached session is STATE (its resident feature matrices are swapped per call) and ctypes releases
        the GIL during native calls, so its use is serialised by a per-model lock: the reference's predict is
        read-only on the model and may be called from several threads (joblib-threaded evaluation, serving).
        A caller that finds the session busy does not wait: it scores on a one-shot session of its own."""
        itf, usf = CSRMatrix(item_features), CSRMatrix(user_features)
        lock = self._scoring_lock()
        if options.cache_scoring_session and lock.acquire(False):
            try:
                sig = tuple(self._array_signature(getattr(self, name)) for name in self._SCORED)
                cached = self.__dict__.get("_scoring")
                if cached is not None and cached[1] != sig:
                    self.__dict__.pop("_scoring", None)
                    cached[0].close()
                    cached = None
                if cached is None:
                    cached = (_Session(self._get_lightfm_data(), itf, usf, scoring=True), sig)
                    self.__dict__["_scoring"] = cached
                else:
                    cached[0].set_features(itf, usf)
                self.__dict__["_scoring_owner"] = threading.get_ident()
                yield cached[0]
            finally:
                self.__dict__.pop("_scoring_owner", None)
                if self.__dict__.pop("_scoring_drop_pending", False):
                    stale = self.__dict__.pop("_scoring", None)
                    if stale is not None:
                        stale[0].close()
                lock.release()
            return
        if not options.cache_scoring_session and lock.acquire(False):
            try:  # caching was switched off after a session had been kept: its device memory goes back now
                stale = self.__dict__.pop("_scoring", None)
                if stale is not None:
                    stale[0].close()
            finally:
                lock.release()
        session = _Session(self._get_lightfm_data(), itf, usf, scoring=True)
        try:
            yield session
        finally:
            session.close()

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_scoring", None)  # a device handle
        state.pop("_scoring_mutex", None)
        state.pop("_scoring_owner", None)
        state.pop("_scoring_drop_pending", None)
        state.pop("_scanned", None)  # per-call scratch of fit_partial
        state.pop("_stages", None)
        return state

    def __del__(self):
        try:
            self._drop_scoring_session()
        except Exception:
            pass

    def _check_initialized(self):
        if any(getattr(self, name) is None for name in _WEIGHTS):
            raise ValueError("You must fit the model before trying to obtain predictions.")

    def _initialize(self, no_components, no_item_features, no_user_features):
        """LFM:281-312 -- item table drawn first, then the user table."""
        start = np.ones if self.learning_schedule == "adagrad" else np.zeros  # accumulators: 1 (adagrad) / 0
        for side, rows in (("item", no_item_features), ("user", no_user_features)):
            # ((rand(rows, d) - 0.5) / d).astype(float32) on the model's RandomState stream, natively (N.init_table)
            setattr(self, side + "_embeddings", N.init_table(self.random_state, rows, no_components))
            setattr(self, side + "_embedding_gradients", start((rows, no_components), dtype=np.float32))
            setattr(self, side + "_embedding_momentum", np.zeros((rows, no_components), dtype=np.float32))
            setattr(self, side + "_biases", np.zeros(rows, dtype=np.float32))
            setattr(self, side + "_bias_gradients", start(rows, dtype=np.float32))
            setattr(self, side + "_bias_momentum", np.zeros(rows, dtype=np.float32))

    def _construct_feature_matrices(self, n_users, n_items, user_features, item_features):
        """LFM:314-363."""
        if user_features is None:
            user_features = sp.identity(n_users, dtype=CYTHON_DTYPE, format="csr")
        else:
            user_features = user_features.tocsr()
        if item_features is None:
            item_features = sp.identity(n_items, dtype=CYTHON_DTYPE, format="csr")
        else:
            item_features = item_features.tocsr()

        if n_users > user_features.shape[0]:
            raise Exception("Number of user feature rows does not equal the number of users")
        if n_items > item_features.shape[0]:
            raise Exception("Number of item feature rows does not equal the number of items")

        # with existing embeddings, every supplied feature must have one
        if self.user_embeddings is not None:
            if not self.user_embeddings.shape[0] >= user_features.shape[1]:
                raise ValueError(
                    "The user feature matrix specifies more features than there are estimated "
                    "feature embeddings: {} vs {}.".format(self.user_embeddings.shape[0],
                                                           user_features.shape[1]))
        if self.item_embeddings is not None:
            if not self.item_embeddings.shape[0] >= item_features.shape[1]:
                raise ValueError(
                    "The item feature matrix specifies more features than there are estimated "
                    "feature embeddings: {} vs {}.".format(self.item_embeddings.shape[0],
                                                           item_features.shape[1]))

        return self._to_cython_dtype(user_features), self._to_cython_dtype(item_features)

    def _get_positives_lookup_matrix(self, interactions):
        """LFM:365-372."""
        mat = interactions.tocsr()
        if not mat.has_sorted_indices:
            return mat.sorted_indices()
        return mat

    def _to_cython_dtype(self, mat):
        if mat.dtype != CYTHON_DTYPE:
            return mat.astype(CYTHON_DTYPE)
        return mat

    def _process_sample_weight(self, interactions, sample_weight):
        """LFM:381-420."""
        if sample_weight is None:
            if self._scan(interactions.data)[0]:   # np.array_equiv(data, 1.0), LFM:383-386
                return interactions.data  # aliases Y, like the reference
            return np.ones_like(interactions.data, dtype=CYTHON_DTYPE)

        if self.loss == "warp-kos":
            raise NotImplementedError("k-OS loss with sample weights not implemented.")
        if not isinstance(sample_weight, sp.coo_matrix):
            raise ValueError("Sample_weight must be a COO matrix.")
        if sample_weight.shape != interactions.shape:
            raise ValueError("Sample weight and interactions matrices must be the same shape")
        if not (np.array_equal(interactions.row, sample_weight.row)
                and np.array_equal(interactions.col, sample_weight.col)):
            raise ValueError("Sample weight and interaction matrix entries must be in the same order")
        if sample_weight.data.dtype != CYTHON_DTYPE:
            return sample_weight.data.astype(CYTHON_DTYPE)
        return sample_weight.data

    def _get_lightfm_data(self):
        """LFM:422-445."""
        return FastLightFM(*[getattr(self, name) for name in _WEIGHTS], self.no_components,
                           int(self.learning_schedule == "adadelta"), self.learning_rate, self.rho,
                           self.epsilon, self.max_sampled)

    def _check_finite(self):
        """LFM:447-464 (host version; fit_partial uses the on-device check)."""
        for parameter in (self.item_embeddings, self.item_biases, self.user_embeddings,
                          self.user_biases):
            if not np.isfinite(np.sum(parameter)):
                raise ValueError(_NOT_FINITE)

    def _scan(self, data):
        """(all ones, finite) of an input array: one pass of the native helper, remembered per array for the call (the
        interaction values are asked about three times: LFM:383-386, 617-625)."""
        seen = getattr(self, "_scanned", None)
        if seen is not None and id(data) in seen and seen[id(data)][0] is data:
            return seen[id(data)][1]
        answer = N.host_scan(data) if isinstance(data, np.ndarray) and data.size >= (1 << 16) else (
            bool(np.array_equiv(data, 1.0)), bool(np.isfinite(np.sum(data))))
        if seen is not None:  # only inside fit_partial, which clears it again: the entries hold the caller's arrays
            seen[id(data)] = (data, answer)
        return answer

    def _check_input_finite(self, data):
        if not self._scan(data)[1]:
            raise ValueError("Not all input values are finite. "
                             "Check the input for NaNs and infinite values.")

    @staticmethod
    def _progress(n, verbose):
        """The epoch counter of fit_partial: silent, a tqdm bar where tqdm is importable, otherwise one printed
        line per epoch (what LFM:474-492 offers)."""
        if verbose:
            try:
                import tqdm
            except ImportError:
                return _printed_epochs(n)
            return tqdm.trange(n, desc="Epoch")
        return range(n)

    # -------------------------------------------------------------------- fit

    def fit(self, interactions, user_features=None, item_features=None, sample_weight=None,
            epochs=1, num_threads=1, verbose=False):
        """Fit from scratch (LFM:494-558).  Arguments as in the reference; `num_threads`
        only sets how many seeds are drawn from `random_state` per epoch."""
        self._reset_state()
        return self.fit_partial(interactions, user_features=user_features,
                                item_features=item_features, sample_weight=sample_weight,
                                epochs=epochs, num_threads=num_threads, verbose=verbose)

    def fit_partial(self, interactions, user_features=None, item_features=None,
                    sample_weight=None, epochs=1, num_threads=1, verbose=False):
        """Resume training from the current state (LFM:560-666)."""
        self._stages = stages = _Stages()
        self._scanned = {}  # per-array answers of _scan for THIS call (feature values, interaction values, sample weights)
        try:
            interactions, user_features, item_features, sample_weight_data = self._fit_prologue(
                interactions, user_features, item_features, sample_weight, num_threads)
        finally:
            self._scanned = None  # (the entries hold references to the caller's arrays)
        stages.mark("host checks")
        self._run_epochs(item_features, user_features, interactions, sample_weight_data,
                         num_threads, epochs, verbose)
        stages.report()
        return self

    def _fit_prologue(self, interactions, user_features, item_features, sample_weight, num_threads):
        """The input coercions and checks of LFM:560-652."""
        interactions = interactions.tocoo()
        if interactions.dtype != CYTHON_DTYPE:
            interactions.data = interactions.data.astype(CYTHON_DTYPE)
        sample_weight_data = self._process_sample_weight(interactions, sample_weight)

        n_users, n_items = interactions.shape
        user_features, item_features = self._construct_feature_matrices(
            n_users, n_items, user_features, item_features)

        for input_data in (user_features.data, item_features.data, interactions.data,
                           sample_weight_data):
            self._check_input_finite(input_data)

        if self.item_embeddings is None:
            self._initialize(self.no_components, item_features.shape[1], user_features.shape[1])

        if not item_features.shape[1] == self.item_embeddings.shape[0]:
            raise ValueError("Incorrect number of features in item_features")
        if not user_features.shape[1] == self.user_embeddings.shape[0]:
            raise ValueError("Incorrect number of features in user_features")
        if num_threads < 1:
            raise ValueError("Number of threads must be 1 or larger.")
        return interactions, user_features, item_features, sample_weight_data

    def _run_epochs(self, item_features, user_features, interactions, sample_weight, num_threads,
                    epochs, verbose):
        """The epoch loop of LFM:654-664 + _run_epoch (LFM:668-759) on one device session."""
        if epochs <= 0:
            return
        self._drop_scoring_session()  # its tables are about to be stale; hand the memory to the fit
        loss = self.loss
        needs_lookup = loss in ("warp", "bpr", "warp-kos")
        positives = None
        if needs_lookup and options.host_positives:
            # the reference's host-side lookup matrix (LFM:365-372), built before the shuffle
            # indices as in LFM:682-690
            positives = CSRMatrix(self._get_positives_lookup_matrix(interactions))
        rows = np.ascontiguousarray(interactions.row, dtype=np.int32)
        cols = np.ascontiguousarray(interactions.col, dtype=np.int32)
        data = np.ascontiguousarray(interactions.data, dtype=np.float32)
        if sample_weight is not interactions.data:
            sample_weight = np.ascontiguousarray(sample_weight, dtype=np.float32)
        else:
            sample_weight = data
        n = len(data)
        device_shuffle = (options.device_shuffle and options.mode == "parallel"
                          and not getattr(self, "_shared_random_state", False))

        stages = getattr(self, "_stages", None) or _Stages()
        model = self._get_lightfm_data()
        session = _Session(model, CSRMatrix(item_features), CSRMatrix(user_features))
        stages.mark("session + tables")
        try:
            if loss == "warp-kos" and positives is not None:
                session.set_interactions(positives, rows, None, None, None)
            else:
                session.set_interactions(positives, rows, cols, data, sample_weight)
            stages.mark("interactions upload")
            if needs_lookup and positives is None:
                # the same matrix (sorted rows, duplicates merged) built on the device from the COO
                session.build_positives(interactions.shape[0], interactions.shape[1])
            stages.mark("positives")
            self._last_epoch_stats = []
            ahead = device_shuffle and options.shuffle_ahead and epochs > 1
            slot, next_keys = 0, None
            for e in self._progress(epochs, verbose=verbose):
                if device_shuffle:
                    # two draws key a permutation built on the device (the caller's RandomState
                    # still advances every epoch, tests/test_movielens.py:669-682 of the reference)
                    # (624 draws = one full Mersenne-Twister block, so get_state()[1] changes too)
                    if next_keys is None:
                        keys = self.random_state.randint(0, np.iinfo(np.int32).max, size=624)
                        session.device_shuffle(int(keys[0]), int(keys[1]), slot=slot)
                else:
                    shuffle_indices = np.arange(n, dtype=np.int32)
                    self.random_state.shuffle(shuffle_indices)
                    session.upload_shuffle(shuffle_indices)
                seeds = None
                if loss != "logistic":  # _lightfm_fast.pyx.template:812-814
                    seeds = np.ascontiguousarray(self.random_state.randint(
                        0, np.iinfo(np.int32).max, size=num_threads).astype(np.uint32))
                next_keys = None
                if ahead and e + 1 < epochs:
                    # the NEXT epoch's permutation goes to the other slot on a stream of its own while this epoch
                    # trains (the draws keep their order: keys, seeds, keys, seeds, ...)
                    next_keys = self.random_state.randint(0, np.iinfo(np.int32).max, size=624)
                    session.device_shuffle_ahead(int(next_keys[0]), int(next_keys[1]), slot=1 - slot)
                opts, _ = make_opts()
                opts.history = int(getattr(self, "_trained_interactions", 0))
                session.epoch(loss, self.item_alpha, self.user_alpha, self.k, self.n, seeds, opts, slot=slot)
                if next_keys is not None:
                    slot = 1 - slot
                self._trained_interactions = getattr(self, "_trained_interactions", 0) + n
                self._last_epoch_stats.append({"kernel_ms": float(opts.kernel_ms),
                                               "counters": list(opts.counters),
                                               "kernel_used": int(opts.kernel_used),
                                               "tile_ng": int(opts.tile_ng),
                                               "in_flight": int(opts.in_flight),
                                               "launches": int(opts.launches),
                                               "user_store": int(opts.user_store),
                                               "plan_flags": int(opts.plan_flags)})
                if not session.check_finite():  # LFM:664
                    session.sync_to_host(model)
                    raise ValueError(_NOT_FINITE)
                stages.mark("epoch")
            session.sync_to_host(model)
            stages.mark("download")
        finally:
            session.close()
            if options.trim_after_fit:
                N.device_trim()

    # ---------------------------------------------------------------- predict

    def predict(self, user_ids, item_ids, item_features=None, user_features=None, num_threads=1):
        """Scores for (user, item) PAIRS (LFM:761-872)."""
        self._check_initialized()

        user_ids, item_ids = _pair_ids(user_ids, item_ids)
        if num_threads < 1:
            raise ValueError("Number of threads must be 1 or larger.")
        if user_ids.min() < 0 or item_ids.min() < 0:
            raise ValueError("User or item ids cannot be negative. Check your inputs for negative "
                             "numbers or very large numbers that can overflow.")

        n_users = user_ids.max() + 1
        n_items = item_ids.max() + 1
        user_features, item_features = self._construct_feature_matrices(
            n_users, n_items, user_features, item_features)

        predictions = np.empty(len(user_ids), dtype=np.float32)
        # predict_lightfm (PYX:1185-1229) on the resident scoring session
        with self._scoring_session(item_features, user_features) as session:
            session.predict(np.ascontiguousarray(user_ids), np.ascontiguousarray(item_ids), predictions)
        return predictions

    def _check_test_train_intersections(self, test_mat, train_mat):
        if train_mat is not None:
            n_intersections = test_mat.multiply(train_mat).nnz
            if n_intersections:
                raise ValueError(
                    "Test interactions matrix and train interactions matrix share %d "
                    "interactions. This will cause incorrect evaluation, check your data split."
                    % n_intersections)

    def predict_rank(self, test_interactions, train_interactions=None, item_features=None,
                     user_features=None, num_threads=1, check_intersections=True):
        """Rank of every test interaction among all items, 0 = best (LFM:884-989)."""
        self._check_initialized()
        if num_threads < 1:
            raise ValueError("Number of threads must be 1 or larger.")
        if check_intersections:
            self._check_test_train_intersections(test_interactions, train_interactions)

        n_users, n_items = test_interactions.shape
        user_features, item_features = self._construct_feature_matrices(
            n_users, n_items, user_features, item_features)
        if not item_features.shape[1] == self.item_embeddings.shape[0]:
            raise ValueError("Incorrect number of features in item_features")
        if not user_features.shape[1] == self.user_embeddings.shape[0]:
            raise ValueError("Incorrect number of features in user_features")

        test_interactions = self._to_cython_dtype(test_interactions.tocsr())
        if train_interactions is None:
            train_interactions = sp.csr_matrix((n_users, n_items), dtype=CYTHON_DTYPE)
        else:
            train_interactions = self._to_cython_dtype(train_interactions.tocsr())

        if not train_interactions.has_sorted_indices:
            # the train-positive mask is a sorted-row lookup (PYX:1303-1304: in_positives' binary search)
            train_interactions = train_interactions.sorted_indices()

        ranks = sp.csr_matrix((np.zeros_like(test_interactions.data), test_interactions.indices,
                               test_interactions.indptr), shape=test_interactions.shape)
        # predict_ranks (PYX:1232-1323) on the resident scoring session
        with self._scoring_session(item_features, user_features) as session:
            session.predict_ranks(CSRMatrix(test_interactions), CSRMatrix(train_interactions), ranks.data)
        return ranks

    # -------------------------------------------------------- representations

    def _representations(self, side, features):
        """features @ (biases, embeddings) on the device: one wavefront per feature row gathers the
        embedding rows (csrc/predict_kernels.hip: rep_rows_kernel; float32, CSR order)."""
        features = sp.csr_matrix(features, dtype=CYTHON_DTYPE)
        table = self.item_embeddings if side == 0 else self.user_embeddings
        if features.shape[1] != table.shape[0]:  # what scipy's `features * embeddings` raises
            raise ValueError("dimension mismatch")
        empty = sp.csr_matrix((0, 0), dtype=CYTHON_DTYPE)
        with self._scoring_session(empty, empty) as session:
            return session.representations(side, CSRMatrix(features))

    def get_item_representations(self, features=None):
        """(biases, embeddings) of items, optionally through a feature matrix (LFM:991-1018)."""
        self._check_initialized()
        if features is None:
            return self.item_biases, self.item_embeddings
        return self._representations(0, features)

    def get_user_representations(self, features=None):
        """(biases, embeddings) of users (LFM:1020-1047)."""
        self._check_initialized()
        if features is None:
            return self.user_biases, self.user_embeddings
        return self._representations(1, features)

    # ---------------------------------------------------------------- sklearn

    def get_params(self, deep=True):
        """LFM:1049-1080."""
        return {"loss": self.loss, "learning_schedule": self.learning_schedule,
                "no_components": self.no_components, "learning_rate": self.learning_rate,
                "k": self.k, "n": self.n, "rho": self.rho, "epsilon": self.epsilon,
                "max_sampled": self.max_sampled, "item_alpha": self.item_alpha,
                "user_alpha": self.user_alpha, "random_state": self.random_state}

    def set_params(self, **params):
        """LFM:1082-1107."""
        valid_params = self.get_params()
        for key, value in params.items():
            if key not in valid_params:
                raise ValueError(
                    "Invalid parameter %s for estimator %s. Check the list of available "
                    "parameters with `estimator.get_params().keys()`."
                    % (key, self.__class__.__name__))
            setattr(self, key, value)
        return self
