"""The reference's OWN host layer and test-suite on top of the HIP backend, through the shim of
INTEGRATION.md section 1 ("tests re-used as-is against the new backend").

oracle/ref_suite.py lays out, in a scratch directory outside the repository, the reference's
`lightfm/{__init__,lightfm,evaluation,data,cross_validation,version}` + the three-branch
`_lightfm_fast.py` + the reference's tests (from /root/reference here; from their bytecode under
oracle/_ref/pysuite on the GPU box, where /root/reference does not exist) and runs pytest on them in a
subprocess:

  * CPU (`-m "not gpu"`): with the reference's own compiled extension -- all 64 tests pass, which pins the
    harness and the synthetic stand-in for `fetch_movielens` (no network: data of the ML-100k shape and
    format, tuned so that the reference clears every floor its tests pin on the real dataset);
  * GPU (`-m gpu`): with LIGHTFM_BACKEND=hip.  `test_api.py` (18, incl. the predict_rank known answers,
    tests/test_api.py:217-282), `test_evaluation.py` (5, the metrics against slow Python
    re-implementations, :164-269), `test_fast_functions.py` (1), `test_data.py` (4),
    `test_cross_validation.py` (3) and `test_movielens.py` (33: every loss, both schedules, regularisation,
    pickling, resuming, sample weights, representations, sklearn CV) must pass in the DEFAULT parallel
    mode -- except the tests that assert run-to-run bit-reproducibility of a single-threaded fit, which
    a Hogwild GPU run cannot give: those (DETERMINISM) are run in serial mode (LIGHTFM_AMD_MODE=serial,
    the bit-exact one-wavefront mode) instead.
"""
import json
import os

import pytest

from oracle import oracle, ref_suite

# single-threaded fits compared bit for bit between two runs (test_movielens.py:655-666): only the
# serial mode reproduces a run; everything else runs in the default parallel mode
DETERMINISM = ("test_random_state_fixing",)
# asserts that hold only without exact score ties, on unseeded models
TIE_SENSITIVE = ("test_predict_ranks",)


def _skip_unless_available():
    if not ref_suite.available():
        pytest.skip("neither /root/reference nor oracle/_ref/pysuite is present")
    if not oracle.ref_available("strict"):
        pytest.skip("oracle/_ref/strict not built")


def _retry_tie_sensitive(dst, backend, outcomes):
    """The reference's predict_rank test asserts that a dense row's ranks are a permutation, on an UNSEEDED model: two
    items with exactly equal float32 scores repeat a (pessimistic) rank.  2.5 % of such models have a pair like that
    -- 76 rows in 3 000 models on the HIP backend, 77 in 3 000 with the reference's own extension
    (profiles/r04_visit_p.txt).  One more model for that test before it counts as failed."""
    retried, tail = {}, ""
    for test_id, outcome in list(outcomes.items()):
        if outcome == "FAILED" and any(t in test_id for t in TIE_SENSITIVE):
            again, _, tail_again = ref_suite.run(dst, backend, files=("test_api",), select=" or ".join(TIE_SENSITIVE))
            retried[test_id] = [outcome, again.get(test_id)]
            if again.get(test_id) == "PASSED":
                outcomes[test_id] = "PASSED"
            tail += tail_again[-1500:]
    return retried, tail


def test_reference_suite_passes_on_the_reference_backend(tmp_path_factory):
    """Harness + synthetic MovieLens stand-in, checked against the reference's own extension."""
    _skip_unless_available()
    dst = str(tmp_path_factory.mktemp("ref_suite_cpu"))
    ref_suite.materialise(dst)
    fast_mod, class_mod, _ = ref_suite.loaded_backend(dst, "reference")
    assert fast_mod.endswith("_lightfm_fast_openmp") and class_mod.endswith("_lightfm_fast_openmp")
    outcomes, rc, tail = ref_suite.run(dst, "reference")
    retried, _ = _retry_tie_sensitive(dst, "reference", outcomes)
    bad = {k: v for k, v in outcomes.items() if v != "PASSED"}
    assert (rc == 0 or retried) and not bad, (bad, tail[-3000:])
    assert len(outcomes) == 64, len(outcomes)


def test_bytecode_bundle_is_equivalent_to_the_sources(tmp_path_factory):
    """What travels to the GPU box (oracle/_ref/pysuite) collects the same tests as the sources."""
    _skip_unless_available()
    if not (os.path.isdir(ref_suite.REF) and os.path.exists(os.path.join(ref_suite.PYSUITE, "tests", "test_api.pyc"))):
        pytest.skip("needs both /root/reference and the built bundle")
    a = str(tmp_path_factory.mktemp("ref_suite_src"))
    b = str(tmp_path_factory.mktemp("ref_suite_pyc"))
    ref_suite.materialise(a)
    saved = ref_suite.REF
    ref_suite.REF = "/nonexistent"
    try:
        ref_suite.materialise(b)
    finally:
        ref_suite.REF = saved
    assert os.path.exists(os.path.join(b, "lightfm", "lightfm.pyc")) and not os.path.exists(os.path.join(b, "lightfm", "lightfm.py"))
    files = ("test_api", "test_evaluation", "test_fast_functions", "test_data")
    oa, rca, _ = ref_suite.run(a, "reference", files=files)
    ob, rcb, tail = ref_suite.run(b, "reference", files=files)
    assert rca == 0 and rcb == 0, tail[-2000:]
    assert oa == ob and len(oa) == 28


@pytest.mark.gpu
def test_reference_suite_on_the_hip_backend(tmp_path_factory):
    from lightfm_amd import _native
    assert _native.device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    _skip_unless_available()
    dst = str(tmp_path_factory.mktemp("ref_suite_hip"))
    ref_suite.materialise(dst)
    fast_mod, class_mod, class_file = ref_suite.loaded_backend(dst, "hip")
    assert fast_mod == "lightfm_amd._lightfm_fast" and class_mod == "lightfm_amd._lightfm_fast", (fast_mod, class_mod)
    assert class_file.startswith(dst)  # the reference's LightFM class, not lightfm_amd's

    deselect = " and ".join("not " + t for t in DETERMINISM)
    outcomes, rc, tail = ref_suite.run(dst, "hip", select=deselect)
    serial, rc2, tail2 = ref_suite.run(dst, "hip", files=("test_movielens",), mode="serial", select=" or ".join(DETERMINISM))
    outcomes.update(serial)
    retried, tail_again = _retry_tie_sensitive(dst, "hip", outcomes)
    if retried and all(v == "PASSED" for k, v in outcomes.items() if k not in serial):
        rc = 0
    tail += tail_again
    record = os.path.join(ref_suite.ROOT, "gpurun_out")
    if os.path.isdir(record):  # evidence for profiles/: which reference tests ran and how they ended
        with open(os.path.join(record, "reference_suite_on_hip.json"), "w") as f:
            json.dump({"outcomes": outcomes, "retried": retried, "tail": tail[-4000:], "tail_serial": tail2[-1500:]}, f, indent=1)
    bad = {k: v for k, v in outcomes.items() if v != "PASSED"}
    assert not bad, (bad, tail[-5000:], tail2[-2000:])
    assert rc == 0 and rc2 == 0
    assert len(outcomes) == 64, len(outcomes)
