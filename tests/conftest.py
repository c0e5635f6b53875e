import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ref_strict():
    from oracle import oracle
    oracle.build()
    if not oracle.ref_available("strict"):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return oracle.ref_module("strict")


@pytest.fixture(autouse=True)
def _restore_backend_options():
    """Backend options a test sets (mode, kernel selectors, ...) never leak into the next test."""
    try:
        from lightfm_amd.options import options
    except Exception:
        yield
        return
    saved = dict(options.__dict__)
    yield
    options.__dict__.clear()
    options.__dict__.update(saved)
