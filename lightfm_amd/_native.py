"""ctypes binding of liblfm_hip.so (C ABI: include/lfm_hip.h).

There is deliberately no fallback: if the HIP library is missing, or no HIP
device is usable, every compute entry point raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# LIGHTFM_AMD_LIB: another build of the library (A/B measurements of kernel variants, tools/ab_build.sh)
LIB_PATH = os.environ.get("LIGHTFM_AMD_LIB") or os.path.join(HERE, "_lib", "liblfm_hip.so")

F32P = C.POINTER(C.c_float)
I32P = C.POINTER(C.c_int32)
U32P = C.POINTER(C.c_uint32)

LOSS_IDS = {"logistic": 0, "warp": 1, "bpr": 2, "warp-kos": 3}
MODE_PARALLEL, MODE_SERIAL = 0, 1
UNIQUE_ID_BYTES = 128


class HipBackendError(RuntimeError):
    pass


class LfmCSR(C.Structure):
    _fields_ = [("indices", I32P), ("indptr", I32P), ("data", F32P),
                ("rows", C.c_int32), ("cols", C.c_int32), ("nnz", C.c_int64)]


MODEL_ARRAYS = ("item_W", "item_G", "item_M", "item_b", "item_bG", "item_bM",
                "user_W", "user_G", "user_M", "user_b", "user_bG", "user_bM")


class LfmModel(C.Structure):
    _fields_ = ([(n, F32P) for n in MODEL_ARRAYS] + [
        ("n_item_feat", C.c_int32), ("n_user_feat", C.c_int32), ("d", C.c_int32),
        ("adadelta", C.c_int32), ("lr", C.c_float), ("rho", C.c_float), ("eps", C.c_float),
        ("max_sampled", C.c_int32), ("item_scale", C.c_double), ("user_scale", C.c_double)])


class LfmOpts(C.Structure):
    _fields_ = [("mode", C.c_int32), ("launches_per_epoch", C.c_int32),
                ("first_batch", C.c_int32), ("max_waves", C.c_int32),
                ("neg_log", I32P), ("sampled_log", I32P),
                ("counters", C.c_int64 * 4), ("kernel_ms", C.c_float),
                ("update_mode", C.c_int32), ("feat_kernel", C.c_int32), ("warp_kernel", C.c_int32),
                ("debug", C.c_int32),
                ("phase_cycles", C.c_int64 * 8), ("tile_ng", C.c_int32), ("in_flight", C.c_int32),
                ("history", C.c_int64), ("ramp_k", C.c_int32), ("launches", C.c_int32),
                ("kernel_used", C.c_int32), ("shared_cap", C.c_int32),
                ("pos_begin", C.c_int64), ("pos_end", C.c_int64),
                ("streams_used", C.c_int32), ("user_store", C.c_int32), ("tile_ahead", C.c_int32),
                ("plan_flags", C.c_int32)]


class LfmItemExport(C.Structure):
    """include/lfm_hip.h: lfm_item_export -- the HIP IPC handles of a session's item-side allocations."""
    _fields_ = [("handle", (C.c_char * 64) * 4), ("offset", C.c_int64 * 4), ("bytes", C.c_int64 * 4),
                ("n_items", C.c_int32), ("d", C.c_int32), ("device", C.c_int32), ("reserved", C.c_int32),
                ("pid", C.c_int64)]


# every symbol include/lfm_hip.h declares (tests check the .so exports them all)
EXPORTS = (
    "lfm_last_error", "lfm_last_kernel_ms", "lfm_device_count", "lfm_device_info",
    "lfm_fit_warp", "lfm_fit_bpr", "lfm_fit_logistic", "lfm_fit_warp_kos",
    "lfm_predict", "lfm_predict_ranks", "lfm_auc_from_rank", "lfm_in_positives",
    "lfm_session_create", "lfm_session_create_scoring", "lfm_session_set_features", "lfm_session_set_interactions", "lfm_session_upload_shuffle",
    "lfm_session_device_shuffle", "lfm_session_device_shuffle_ahead", "lfm_shuffle_permutation", "lfm_session_download_shuffle",
    "lfm_session_epoch", "lfm_session_check_finite", "lfm_session_predict",
    "lfm_session_predict_ranks", "lfm_session_sync_to_host", "lfm_session_load_model",
    "lfm_session_build_positives", "lfm_session_download_positives", "lfm_session_representations",
    "lfm_session_destroy",
    "lfm_device_trim", "lfm_device_pool_stats", "lfm_selftest_adagrad_cell", "lfm_selftest_ranks_bf16_band", "lfm_host_scan_f32", "lfm_host_mt19937_table", "lfm_host_checksum_u32",
    "lfm_comm_preload", "lfm_comm_unique_id", "lfm_session_comm_init", "lfm_session_comm_merge", "lfm_session_comm_merge_sparse", "lfm_session_comm_merge_flush", "lfm_session_set_merge_dense_fraction",
    "lfm_sessions_merge_local_sparse", "lfm_sessions_merge_local_flush", "lfm_session_merge_begin",
    "lfm_session_set_hot_rows", "lfm_session_comm_merge_hot", "lfm_sessions_merge_local_hot",
    "lfm_session_comm_any", "lfm_session_comm_barrier", "lfm_sessions_merge_local",
    "lfm_sessions_share_items_local",
    "lfm_session_export_items", "lfm_session_share_items_ipc", "lfm_session_gather_shared_items",
)
MERGE_SUM, MERGE_MEAN, MERGE_ADAGRAD = 0, 1, 2
MERGE_MODES = {"sum": MERGE_SUM, "mean": MERGE_MEAN, "adagrad": MERGE_ADAGRAD}

_lib = None


def lib():
    """Loads liblfm_hip.so; raises HipBackendError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipBackendError(
                "%s not found: build it with `python -m lightfm_amd.build` "
                "(the HIP backend has no CPU fallback)" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        l.lfm_last_error.restype = C.c_char_p
        l.lfm_last_kernel_ms.restype = C.c_float
        for name in EXPORTS:
            if name not in ("lfm_last_error", "lfm_last_kernel_ms"):
                getattr(l, name).restype = C.c_int
        _lib = l
    return _lib


def check(rc):
    """Maps LFM_E* codes to the exception types the reference raises."""
    if rc >= 0:
        return rc
    msg = lib().lfm_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg)
    if rc == -3:
        raise MemoryError(msg)
    if rc == -5:
        raise NotImplementedError(msg)
    raise HipBackendError(msg)


def f32p(a):
    return a.ctypes.data_as(F32P) if a is not None else None


def i32p(a):
    return a.ctypes.data_as(I32P) if a is not None else None


def u32p(a):
    return a.ctypes.data_as(U32P) if a is not None else None


def require(a, dtype, ndim, name):
    """The typed-memoryview contract of the Cython signatures (`flt[::1]`, `int[::1]`)."""
    if not isinstance(a, np.ndarray):
        raise TypeError("%s must be a numpy array" % name)
    if a.dtype != dtype:
        raise ValueError("Buffer dtype mismatch, expected '%s' but got '%s' (%s)"
                         % (np.dtype(dtype).name, a.dtype.name, name))
    if a.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions (expected %d, got %d) (%s)"
                         % (ndim, a.ndim, name))
    if not a.flags.c_contiguous:
        raise ValueError("ndarray is not C-contiguous (%s)" % name)
    return a


def host_scan(a):
    """(all values == 1.0, all finite with a finite float32 sum) of a C-contiguous float32 array in one pass
    (lfm_host_scan_f32); numpy when the library has not been built (host-side validation only)."""
    scan = getattr(lib(), "lfm_host_scan_f32", None) if os.path.exists(LIB_PATH) else None
    if scan is not None and a.dtype == np.float32 and a.flags.c_contiguous:
        ones, fin = C.c_int32(), C.c_int32()
        check(scan(f32p(a), C.c_int64(a.size), C.byref(ones), C.byref(fin)))
        return bool(ones.value), bool(fin.value)
    return bool(np.array_equiv(a, 1.0)), bool(np.isfinite(np.sum(a)))


def checksum(a):
    """Position-sensitive exact signature of a C-contiguous float32 array (lfm_host_checksum_u32); a numpy restatement
    of the same sum when the library is not built."""
    flat = a.reshape(-1).view(np.uint32)
    fn = getattr(lib(), "lfm_host_checksum_u32", None) if os.path.exists(LIB_PATH) else None
    if fn is not None:
        out = C.c_uint64()
        check(fn(flat.ctypes.data_as(U32P), C.c_int64(flat.size), C.byref(out)))
        return int(out.value)
    w = np.arange(flat.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    return int(np.add.reduce(flat.astype(np.uint64) * w, dtype=np.uint64))


def init_table(random_state, rows, d):
    """((random_state.rand(rows, d) - 0.5) / d).astype(float32) -- the reference's embedding initialisation
    (LFM:281-312) -- drawn by the native restatement of numpy's MT19937 stream (lfm_host_mt19937_table) on the
    RandomState's own state; numpy itself when the library is not built or the generator is not a legacy MT19937."""
    fill = getattr(lib(), "lfm_host_mt19937_table", None) if os.path.exists(LIB_PATH) else None
    if fill is not None and isinstance(random_state, np.random.RandomState) and rows * d >= (1 << 14):
        st = random_state.get_state()
        if st[0] == "MT19937":
            key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
            pos = C.c_int32(int(st[2]))
            out = np.empty((rows, d), np.float32)
            check(fill(key.ctypes.data_as(U32P), C.byref(pos), f32p(out), C.c_int64(rows * d), C.c_int32(d)))
            random_state.set_state((st[0], key, pos.value, st[3], st[4]))
            return out
    draw = random_state.rand(rows, d)  # float64; the two steps below in place: the same values as
    draw -= 0.5                        # ((rand - 0.5) / d).astype(float32) without two temporaries
    draw /= d
    return draw.astype(np.float32)


def device_trim():
    """Hands the library's cached device memory back to the HIP runtime; returns the bytes released."""
    released = C.c_int64()
    check(lib().lfm_device_trim(C.byref(released)))
    return released.value


def device_pool_stats():
    """(bytes held from the HIP runtime, bytes of them currently unused)."""
    reserved, cached = C.c_int64(), C.c_int64()
    check(lib().lfm_device_pool_stats(C.byref(reserved), C.byref(cached)))
    return reserved.value, cached.value


def preload_comm():
    """Resolve RCCL before anything else in the process (torch) can shadow it; True when it is available."""
    return lib().lfm_comm_preload() == 0


def device_count():
    return lib().lfm_device_count()


def device_info(device=0):
    name = C.create_string_buffer(256)
    cus = C.c_int32()
    mem = C.c_int64()
    check(lib().lfm_device_info(device, name, C.byref(cus), C.byref(mem)))
    return name.value.decode(), cus.value, mem.value
