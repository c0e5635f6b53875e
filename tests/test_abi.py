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


def test_struct_layouts_match_header(native):
    import ctypes as C
    assert C.sizeof(native.LfmCSR) == 40
    assert C.sizeof(native.LfmModel) == 12 * 8 + 4 * 4 + 3 * 4 + 4 + 2 * 8
    assert C.sizeof(native.LfmOpts) == 16 + 16 + 32 + 4 + 4 + 4 + 8 + 4 + 64 + 8 + 16


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
