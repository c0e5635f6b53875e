"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/lfm_hip.h declares, and fails loudly (no CPU fallback) without a GPU."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    from lightfm_amd import build
    build.build()
    from lightfm_amd import _native
    return _native


def test_library_exports_every_declared_symbol(native):
    header = open(os.path.join(ROOT, "include", "lfm_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(lfm_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(native.EXPORTS)
    lib = native.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), name


def test_struct_layouts_match_header(native, tmp_path):
    """The ctypes mirror has the layout gcc gives include/lfm_hip.h (sizes and every field offset)."""
    import ctypes as C
    import subprocess
    exe = str(tmp_path / "abi_probe")
    subprocess.check_call(["gcc", "-std=c11", "-o", exe, os.path.join(ROOT, "tests", "abi_probe.c")])
    layout = dict(line.split() for line in subprocess.check_output([exe]).decode().splitlines())
    mirror = {"lfm_csr": native.LfmCSR, "lfm_model": native.LfmModel, "lfm_opts": native.LfmOpts}
    checked = 0
    for key, value in layout.items():
        if "." in key:
            struct, field = key.split(".")
            assert getattr(mirror[struct], field).offset == int(value), key
            checked += 1
        else:
            assert C.sizeof(mirror[key]) == int(value), key
    assert checked >= 35
    # every field of the header's structs is mirrored (same count)
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "lfm_hip.h")).read(), flags=re.S)
    body = re.search(r"typedef struct lfm_opts \{(.*?)\} lfm_opts;", header, flags=re.S).group(1)
    n_fields = sum(len(stmt.split(",")) for stmt in body.split(";") if stmt.strip())
    assert n_fields == len(native.LfmOpts._fields_)


def test_no_cpu_fallback_without_gpu(native):
    if native.device_count() > 0:
        pytest.skip("a GPU is present")
    import lightfm_amd._lightfm_fast as fast
    mat = sp.csr_matrix(np.array([[0, 1], [1, 0]], dtype=np.float32))
    with pytest.raises(native.HipBackendError):
        getattr(fast, "__test_in_positives")(0, 1, fast.CSRMatrix(mat))


def test_csrmatrix_rejects_wrong_dtypes(native):
    import lightfm_amd._lightfm_fast as fast
    mat = sp.csr_matrix(np.array([[0, 1], [1, 0]], dtype=np.float64))
    with pytest.raises(ValueError):
        fast.CSRMatrix(mat)


def test_argument_validation_happens_before_the_device_is_touched(native):
    """lfm_session_create rejects malformed models / matrices with LFM_EINVAL (-> ValueError, what
    the reference raises for bad input) even without a GPU; well-formed input then fails with
    the no-device error instead of falling back to the CPU."""
    import ctypes as C
    import lightfm_amd._lightfm_fast as fast
    lib = native.lib()
    d, nu, ni = 8, 5, 7
    arrs = [np.zeros((ni, d), np.float32)] * 3 + [np.zeros(ni, np.float32)] * 3 + \
           [np.zeros((nu, d), np.float32)] * 3 + [np.zeros(nu, np.float32)] * 3
    model = fast.FastLightFM(*arrs, d, 0, 0.05, 0.95, 1e-6, 10)
    itf = fast.CSRMatrix(sp.identity(ni, dtype=np.float32, format="csr"))
    usf = fast.CSRMatrix(sp.identity(nu, dtype=np.float32, format="csr"))
    handle = C.c_void_p()

    def create(m, a, b):
        return lib.lfm_session_create(C.byref(handle), 0, m, a, b)

    assert create(None, itf.byref(), usf.byref()) == -1                      # null model
    assert b"null" in lib.lfm_last_error()
    wide = fast.CSRMatrix(sp.identity(ni + 3, dtype=np.float32, format="csr"))
    assert create(model.byref(), wide.byref(), usf.byref()) == -1            # more columns than embeddings
    bad = fast.CSRMatrix(sp.identity(ni, dtype=np.float32, format="csr"))
    bad.indptr[-1] += 1                                                       # indptr does not span nnz
    assert create(model.byref(), bad.byref(), usf.byref()) == -1
    big = fast.FastLightFM(*[np.zeros((ni, 1100), np.float32)] * 3, *[np.zeros(ni, np.float32)] * 3,
                           *[np.zeros((nu, 1100), np.float32)] * 3, *[np.zeros(nu, np.float32)] * 3,
                           1100, 0, 0.05, 0.95, 1e-6, 10)
    assert create(big.byref(), itf.byref(), usf.byref()) == -5               # LFM_EUNSUPPORTED: d > LFM_MAX_COMPONENTS = 1 024
    with pytest.raises(NotImplementedError):
        native.check(-5)
    with pytest.raises(ValueError):
        native.check(-1)
    if native.device_count() == 0:
        assert create(model.byref(), itf.byref(), usf.byref()) == -2         # LFM_ENODEV, no fallback
        assert b"no CPU fallback" in lib.lfm_last_error()
        with pytest.raises(native.HipBackendError):
            native.check(-2)
    else:
        assert create(model.byref(), itf.byref(), usf.byref()) == 0
        assert lib.lfm_session_destroy(handle) == 0


def test_host_permutation_rejects_bad_sizes(native):
    import ctypes as C
    out = np.empty(4, np.int32)
    assert native.lib().lfm_shuffle_permutation(native.i32p(out), C.c_int64(-1), 1, 2) == -1
    assert native.lib().lfm_shuffle_permutation(None, C.c_int64(4), 1, 2) == -1
