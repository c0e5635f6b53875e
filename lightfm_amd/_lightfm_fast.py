"""Drop-in replacement for the reference's native module `lightfm._lightfm_fast`
(/root/reference/lightfm/_lightfm_fast.pyx.template, "PYX" below): same names,
same positional signatures, results written in place into the caller's numpy
arrays -- but every function runs on the MI355X through liblfm_hip.so.

`lightfm/lightfm.py:8-17` and `lightfm/evaluation.py:9` import exactly these
names, so `import lightfm_amd._lightfm_fast as _lightfm_fast` slots in there.

Execution options that are not part of the reference API are read from
`lightfm_amd.options` (mode "parallel" | "serial", launches_per_epoch, ...).
"""
import ctypes as C

import numpy as np

from . import _native as N
from .options import options

__all__ = ["CSRMatrix", "FastLightFM", "fit_logistic", "fit_warp", "fit_bpr", "fit_warp_kos",
           "predict_lightfm", "predict_ranks", "calculate_auc_from_rank", "__test_in_positives"]


class CSRMatrix(object):
    """PYX:145-182 -- a borrowed view of a scipy CSR matrix (int32 / float32)."""

    def __init__(self, csr_matrix):
        self.indices = N.require(csr_matrix.indices, np.int32, 1, "indices")
        self.indptr = N.require(csr_matrix.indptr, np.int32, 1, "indptr")
        self.data = N.require(csr_matrix.data, np.float32, 1, "data")
        self.rows, self.cols = csr_matrix.shape
        self.nnz = len(self.data)
        self._struct = N.LfmCSR(N.i32p(self.indices), N.i32p(self.indptr), N.f32p(self.data),
                                self.rows, self.cols, self.nnz)

    def byref(self):
        return C.byref(self._struct)


class FastLightFM(object):
    """PYX:185-259 -- the 12 weight arrays + hyper-parameters, constructor order kept."""

    _names = ("item_features", "item_feature_gradients", "item_feature_momentum",
              "item_biases", "item_bias_gradients", "item_bias_momentum",
              "user_features", "user_feature_gradients", "user_feature_momentum",
              "user_biases", "user_bias_gradients", "user_bias_momentum")

    def __init__(self, item_features, item_feature_gradients, item_feature_momentum,
                 item_biases, item_bias_gradients, item_bias_momentum,
                 user_features, user_feature_gradients, user_feature_momentum,
                 user_biases, user_bias_gradients, user_bias_momentum,
                 no_components, adadelta, learning_rate, rho, epsilon, max_sampled):
        arrays = (item_features, item_feature_gradients, item_feature_momentum,
                  item_biases, item_bias_gradients, item_bias_momentum,
                  user_features, user_feature_gradients, user_feature_momentum,
                  user_biases, user_bias_gradients, user_bias_momentum)
        for name, a in zip(self._names, arrays):
            setattr(self, name, N.require(a, np.float32, 2 if "feature" in name else 1, name))
        self.no_components = int(no_components)
        self.adadelta = int(adadelta)
        self.learning_rate = float(learning_rate)
        self.rho = float(rho)
        self.eps = float(epsilon)
        self.max_sampled = int(max_sampled)
        self.item_scale = 1.0
        self.user_scale = 1.0
        for side in ("item", "user"):
            w = getattr(self, side + "_features")
            if w.shape[1] != self.no_components:
                raise ValueError("%s_features has %d columns, no_components is %d"
                                 % (side, w.shape[1], self.no_components))
        self._struct = N.LfmModel(*[N.f32p(getattr(self, n)) for n in self._names],
                                  self.item_features.shape[0], self.user_features.shape[0],
                                  self.no_components, self.adadelta, self.learning_rate, self.rho,
                                  self.eps, self.max_sampled, 1.0, 1.0)

    def byref(self):
        return C.byref(self._struct)


def make_opts(n=0, want_log=False):
    o = N.LfmOpts()
    o.mode = N.MODE_SERIAL if options.mode == "serial" else N.MODE_PARALLEL
    o.launches_per_epoch = int(options.launches_per_epoch)
    o.first_batch = int(options.first_batch)
    o.max_waves = int(options.max_waves)
    o.update_mode = int(options.update_mode)
    o.feat_kernel = int(options.feat_kernel)
    o.warp_kernel = int(options.warp_kernel)
    o.debug = int(options.debug)
    o.ramp_k = int(options.ramp_k)
    o.shared_cap = int(options.shared_cap)
    o.history = int(options.history)
    logs = None
    if want_log:
        logs = (np.full(n, -1, np.int32), np.zeros(n, np.int32))
        o.neg_log, o.sampled_log = N.i32p(logs[0]), N.i32p(logs[1])
    return o, logs


def _record(o, logs):
    options.last_counters = list(o.counters)
    options.last_kernel_ms = float(o.kernel_ms)
    options.last_kernel_used = int(o.kernel_used)
    options.last_launches = int(o.launches)
    options.last_streams_used = int(o.streams_used)
    options.last_tile_ng = int(o.tile_ng)
    options.last_tile_ahead = int(o.tile_ahead)
    options.last_user_store = int(o.user_store)
    options.last_plan_flags = int(o.plan_flags)
    options.last_phase_cycles = list(o.phase_cycles)
    options.last_logs = logs


def _draw_seeds(random_state, num_threads):
    # PYX:812-814 -- consumed from the caller's RandomState exactly like the reference
    return np.ascontiguousarray(
        random_state.randint(0, np.iinfo(np.int32).max, size=num_threads).astype(np.uint32))


def _ids(a, name):
    return N.require(a, np.int32, 1, name)


def fit_warp(item_features, user_features, interactions, user_ids, item_ids, Y, sample_weight,
             shuffle_indices, lightfm, learning_rate, item_alpha, user_alpha, num_threads,
             random_state):
    """PYX:784-912.  `learning_rate` is unused, as in the reference (C_OMP:6968)."""
    seeds = _draw_seeds(random_state, num_threads)
    n = len(N.require(Y, np.float32, 1, "Y"))
    o, logs = make_opts(n, options.log_samples)
    N.check(N.lib().lfm_fit_warp(
        item_features.byref(), user_features.byref(), interactions.byref(),
        N.i32p(_ids(user_ids, "user_ids")), N.i32p(_ids(item_ids, "item_ids")), N.f32p(Y),
        N.f32p(N.require(sample_weight, np.float32, 1, "sample_weight")),
        N.i32p(_ids(shuffle_indices, "shuffle_indices")), C.c_int64(n), lightfm.byref(),
        C.c_double(item_alpha), C.c_double(user_alpha), N.u32p(seeds), len(seeds), C.byref(o)))
    _record(o, logs)


def fit_bpr(item_features, user_features, interactions, user_ids, item_ids, Y, sample_weight,
            shuffle_indices, lightfm, learning_rate, item_alpha, user_alpha, num_threads,
            random_state):
    """PYX:1074-1182."""
    seeds = _draw_seeds(random_state, num_threads)
    n = len(N.require(Y, np.float32, 1, "Y"))
    o, logs = make_opts(n, options.log_samples)
    N.check(N.lib().lfm_fit_bpr(
        item_features.byref(), user_features.byref(), interactions.byref(),
        N.i32p(_ids(user_ids, "user_ids")), N.i32p(_ids(item_ids, "item_ids")), N.f32p(Y),
        N.f32p(N.require(sample_weight, np.float32, 1, "sample_weight")),
        N.i32p(_ids(shuffle_indices, "shuffle_indices")), C.c_int64(n), lightfm.byref(),
        C.c_double(item_alpha), C.c_double(user_alpha), N.u32p(seeds), len(seeds), C.byref(o)))
    _record(o, logs)


def fit_warp_kos(item_features, user_features, data, user_ids, shuffle_indices, lightfm,
                 learning_rate, item_alpha, user_alpha, k, n, num_threads, random_state):
    """PYX:915-1071."""
    seeds = _draw_seeds(random_state, num_threads)
    count = len(_ids(user_ids, "user_ids"))
    o, logs = make_opts(count, options.log_samples)
    N.check(N.lib().lfm_fit_warp_kos(
        item_features.byref(), user_features.byref(), data.byref(), N.i32p(user_ids),
        N.i32p(_ids(shuffle_indices, "shuffle_indices")), C.c_int64(count), lightfm.byref(),
        C.c_double(item_alpha), C.c_double(user_alpha), C.c_int32(k), C.c_int32(n),
        N.u32p(seeds), len(seeds), C.byref(o)))
    _record(o, logs)


def fit_logistic(item_features, user_features, user_ids, item_ids, Y, sample_weight,
                 shuffle_indices, lightfm, learning_rate, item_alpha, user_alpha, num_threads):
    """PYX:694-781."""
    n = len(N.require(Y, np.float32, 1, "Y"))
    o, logs = make_opts()
    N.check(N.lib().lfm_fit_logistic(
        item_features.byref(), user_features.byref(), N.i32p(_ids(user_ids, "user_ids")),
        N.i32p(_ids(item_ids, "item_ids")), N.f32p(Y),
        N.f32p(N.require(sample_weight, np.float32, 1, "sample_weight")),
        N.i32p(_ids(shuffle_indices, "shuffle_indices")), C.c_int64(n), lightfm.byref(),
        C.c_double(item_alpha), C.c_double(user_alpha), C.byref(o)))
    _record(o, logs)


def predict_lightfm(item_features, user_features, user_ids, item_ids, predictions, lightfm,
                    num_threads):
    """PYX:1185-1229."""
    n = len(N.require(predictions, np.float32, 1, "predictions"))
    N.check(N.lib().lfm_predict(
        item_features.byref(), user_features.byref(), N.i32p(_ids(user_ids, "user_ids")),
        N.i32p(_ids(item_ids, "item_ids")), N.f32p(predictions), C.c_int64(n), lightfm.byref()))


def predict_ranks(item_features, user_features, test_interactions, train_interactions, ranks,
                  lightfm, num_threads):
    """PYX:1232-1323."""
    N.require(ranks, np.float32, 1, "ranks")
    N.check(N.lib().lfm_predict_ranks(
        item_features.byref(), user_features.byref(), test_interactions.byref(),
        train_interactions.byref(), N.f32p(ranks), lightfm.byref()))


def calculate_auc_from_rank(ranks, num_train_positives, rank_data, auc, num_threads):
    """PYX:1326-1376."""
    N.check(N.lib().lfm_auc_from_rank(
        ranks.byref(), N.i32p(_ids(num_train_positives, "num_train_positives")),
        N.f32p(N.require(rank_data, np.float32, 1, "rank_data")),
        N.f32p(N.require(auc, np.float32, 1, "auc"))))


def __test_in_positives(row, col, mat):
    """PYX:1380-1385."""
    return bool(N.check(N.lib().lfm_in_positives(C.c_int32(row), C.c_int32(col), mat.byref())))
