"""ctypes front end of the CPU oracle (oracle/lfm_oracle.c) and loader for the
compiled reference (oracle/_ref, built from /root/reference by oracle/Makefile).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (lightfm_amd/) never imports
this module.
"""
import ctypes as C
import importlib.util
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liblfm_oracle.so")
REF_SRC = "/root/reference/lightfm/_lightfm_fast_openmp.c"

F32P = C.POINTER(C.c_float)
I32P = C.POINTER(C.c_int32)
U32P = C.POINTER(C.c_uint32)


class OrcCSR(C.Structure):
    _fields_ = [("indices", I32P), ("indptr", I32P), ("data", F32P),
                ("rows", C.c_int32), ("cols", C.c_int32), ("nnz", C.c_int64)]


class OrcModel(C.Structure):
    _fields_ = ([(n, F32P) for n in (
        "item_W", "item_G", "item_M", "item_b", "item_bG", "item_bM",
        "user_W", "user_G", "user_M", "user_b", "user_bG", "user_bM")] + [
        ("n_item_feat", C.c_int32), ("n_user_feat", C.c_int32), ("d", C.c_int32),
        ("adadelta", C.c_int32), ("lr", C.c_float), ("rho", C.c_float), ("eps", C.c_float),
        ("max_sampled", C.c_int32), ("item_scale", C.c_double), ("user_scale", C.c_double)])


class OrcOpts(C.Structure):
    _fields_ = [("rng_mode", C.c_int32), ("dot_mode", C.c_int32), ("neg_log", I32P),
                ("sampled_log", I32P), ("counters", C.c_int64 * 4)]


def build(force=False):
    """Compile the C restatement (and, where /root/reference exists, oracle/_ref)."""
    if force or not os.path.exists(LIB_PATH) or (
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "lfm_oracle.c"))):
        subprocess.check_call(["make", "-C", HERE, "liblfm_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists(REF_SRC):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-C", HERE, "pysuite"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_rand_r.restype = C.c_int32
        _lib.orc_position_seed.restype = C.c_uint32
        _lib.orc_position_seed.argtypes = [C.c_uint32, C.c_uint64]
        _lib.orc_in_positives.restype = C.c_int32
    return _lib


def ref_available(kind="strict"):
    return os.path.exists(os.path.join(HERE, "_ref", kind, "_lightfm_fast_openmp.so"))


_ref_mods = {}


def ref_module(kind="strict"):
    """The reference's own native extension, compiled by `make -C oracle ref`.

    kind = "strict" (LIGHTFM_NO_CFLAGS build, parity) or "fast" (default-flag
    build, CPU performance baseline).
    """
    if kind not in _ref_mods:
        path = os.path.join(HERE, "_ref", kind, "_lightfm_fast_openmp.so")
        spec = importlib.util.spec_from_file_location("_lightfm_fast_openmp", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _ref_mods[kind] = mod
    return _ref_mods[kind]


def _f32(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(F32P)


def _i32(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(I32P)


def csr_struct(m):
    """scipy csr_matrix (float32 data, int32 indices) -> OrcCSR (borrowing)."""
    s = OrcCSR(_i32(m.indices), _i32(m.indptr), _f32(m.data), m.shape[0], m.shape[1],
               len(m.data))
    s._keep = m
    return s


ARRAYS = ("item_embeddings", "item_embedding_gradients", "item_embedding_momentum",
          "item_biases", "item_bias_gradients", "item_bias_momentum",
          "user_embeddings", "user_embedding_gradients", "user_embedding_momentum",
          "user_biases", "user_bias_gradients", "user_bias_momentum")


class State:
    """The 12 LightFM weight arrays + hyper-parameters (lightfm.py:245-312)."""

    def __init__(self, n_item_feat, n_user_feat, d, rng, schedule="adagrad", lr=0.05, rho=0.95,
                 eps=1e-6, max_sampled=10):
        # same draw order as LightFM._initialize (lightfm.py:281-312): items first
        self.item_embeddings = ((rng.rand(n_item_feat, d) - 0.5) / d).astype(np.float32)
        self.user_embeddings = ((rng.rand(n_user_feat, d) - 0.5) / d).astype(np.float32)
        for side, n in (("item", n_item_feat), ("user", n_user_feat)):
            setattr(self, side + "_embedding_gradients", np.zeros((n, d), np.float32))
            setattr(self, side + "_embedding_momentum", np.zeros((n, d), np.float32))
            setattr(self, side + "_biases", np.zeros(n, np.float32))
            setattr(self, side + "_bias_gradients", np.zeros(n, np.float32))
            setattr(self, side + "_bias_momentum", np.zeros(n, np.float32))
        if schedule == "adagrad":
            for side in ("item", "user"):
                getattr(self, side + "_embedding_gradients")[...] = 1
                getattr(self, side + "_bias_gradients")[...] = 1
        self.d, self.schedule, self.lr, self.rho, self.eps = d, schedule, lr, rho, eps
        self.max_sampled = max_sampled

    def copy(self):
        other = object.__new__(State)
        other.__dict__.update(self.__dict__)
        for n in ARRAYS:
            setattr(other, n, getattr(self, n).copy())
        return other

    def arrays(self):
        return [getattr(self, n) for n in ARRAYS]

    def struct(self):
        m = OrcModel(*[_f32(a) for a in self.arrays()],
                     self.item_embeddings.shape[0], self.user_embeddings.shape[0], self.d,
                     int(self.schedule == "adadelta"), self.lr, self.rho, self.eps,
                     self.max_sampled, 1.0, 1.0)
        m._keep = self
        return m

    def ref_struct(self, mod):
        """FastLightFM of the compiled reference over the same arrays."""
        return mod.FastLightFM(*self.arrays(), self.d, int(self.schedule == "adadelta"), self.lr,
                               self.rho, self.eps, self.max_sampled)


class Opts:
    def __init__(self, n=0, rng_mode=0, dot_mode=0, log=False):
        self.neg = np.full(n, -1, np.int32) if log else None
        self.sampled = np.zeros(n, np.int32) if log else None
        self.c = OrcOpts(rng_mode, dot_mode, _i32(self.neg) if log else None,
                         _i32(self.sampled) if log else None)

    @property
    def counters(self):
        return list(self.c.counters)


def _seeds(seeds):
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    return seeds, seeds.ctypes.data_as(U32P), len(seeds)


def fit_warp(itf, usf, positives, rows, cols, Y, weight, shuffle, state, item_alpha, user_alpha,
             seeds, opts=None):
    s, sp_, ns = _seeds(seeds)
    o = opts or Opts()
    m = state.struct()
    rc = lib().orc_fit_warp(C.byref(csr_struct(itf)), C.byref(csr_struct(usf)),
                            C.byref(csr_struct(positives)), _i32(rows), _i32(cols), _f32(Y),
                            _f32(weight), _i32(shuffle), C.c_int64(len(shuffle)), C.byref(m),
                            C.c_double(item_alpha), C.c_double(user_alpha), sp_, ns, C.byref(o.c))
    assert rc == 0
    return o


def fit_bpr(itf, usf, positives, rows, cols, Y, weight, shuffle, state, item_alpha, user_alpha,
            seeds, opts=None):
    s, sp_, ns = _seeds(seeds)
    o = opts or Opts()
    m = state.struct()
    rc = lib().orc_fit_bpr(C.byref(csr_struct(itf)), C.byref(csr_struct(usf)),
                           C.byref(csr_struct(positives)), _i32(rows), _i32(cols), _f32(Y),
                           _f32(weight), _i32(shuffle), C.c_int64(len(shuffle)), C.byref(m),
                           C.c_double(item_alpha), C.c_double(user_alpha), sp_, ns, C.byref(o.c))
    assert rc == 0
    return o


def fit_logistic(itf, usf, rows, cols, Y, weight, shuffle, state, item_alpha, user_alpha,
                 opts=None):
    o = opts or Opts()
    m = state.struct()
    rc = lib().orc_fit_logistic(C.byref(csr_struct(itf)), C.byref(csr_struct(usf)), _i32(rows),
                                _i32(cols), _f32(Y), _f32(weight), _i32(shuffle),
                                C.c_int64(len(shuffle)), C.byref(m), C.c_double(item_alpha),
                                C.c_double(user_alpha), C.byref(o.c))
    assert rc == 0
    return o


def fit_warp_kos(itf, usf, data, rows, shuffle, state, item_alpha, user_alpha, k, n, seeds,
                 opts=None):
    s, sp_, ns = _seeds(seeds)
    o = opts or Opts()
    m = state.struct()
    rc = lib().orc_fit_warp_kos(C.byref(csr_struct(itf)), C.byref(csr_struct(usf)),
                                C.byref(csr_struct(data)), _i32(rows), _i32(shuffle),
                                C.c_int64(len(shuffle)), C.byref(m), C.c_double(item_alpha),
                                C.c_double(user_alpha), C.c_int32(k), C.c_int32(n), sp_, ns,
                                C.byref(o.c))
    assert rc == 0
    return o


def predict(itf, usf, uids, iids, state, dot_mode=0):
    out = np.empty(len(uids), np.float32)
    m = state.struct()
    rc = lib().orc_predict(C.byref(csr_struct(itf)), C.byref(csr_struct(usf)), _i32(uids),
                           _i32(iids), _f32(out), C.c_int64(len(uids)), C.byref(m),
                           C.c_int32(dot_mode))
    assert rc == 0
    return out


def predict_ranks(itf, usf, test, train, ranks, state, dot_mode=0):
    m = state.struct()
    rc = lib().orc_predict_ranks(C.byref(csr_struct(itf)), C.byref(csr_struct(usf)),
                                 C.byref(csr_struct(test)), C.byref(csr_struct(train)),
                                 _f32(ranks), C.byref(m), C.c_int32(dot_mode))
    assert rc == 0


def auc_from_rank(ranks_csr, num_train_positives, rank_data, auc):
    rc = lib().orc_auc_from_rank(C.byref(csr_struct(ranks_csr)), _i32(num_train_positives),
                                 _f32(rank_data), _f32(auc))
    assert rc == 0


def rand_r_stream(seed, n):
    st = C.c_uint32(seed)
    return np.array([lib().orc_rand_r(C.byref(st)) for _ in range(n)], np.int64), st.value


def position_seed(base, i):
    return lib().orc_position_seed(C.c_uint32(base), C.c_uint64(i))


def in_positives(item, user, m):
    return bool(lib().orc_in_positives(C.c_int32(item), C.c_int32(user), C.byref(csr_struct(m))))
