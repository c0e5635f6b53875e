"""Runs the REFERENCE's own host layer and test-suite against a chosen native backend.

TEST INFRASTRUCTURE ONLY (see oracle/lfm_oracle.c): used by tests/test_reference_suite.py and
tests/golden/make_ref_suite_outcomes.py.  Nothing in lightfm_amd/ imports this.

What it does: builds, in a scratch directory OUTSIDE the repository, the package a maintainer of
lyst/lightfm would have after following INTEGRATION.md section 1 --

    <scratch>/lightfm/{__init__,lightfm,evaluation,data,cross_validation,version}   the reference's files
    <scratch>/lightfm/_lightfm_fast.py                 INTEGRATION.md's three-branch shim (written here)
    <scratch>/lightfm/_lightfm_fast_openmp.so          the reference's compiled extension (oracle/_ref/strict)
    <scratch>/lightfm/datasets/__init__.py             a SYNTHETIC fetch_movielens (written here; no network)
    <scratch>/tests/test_*.py                          the reference's tests

-- and runs pytest on it in a subprocess with LIGHTFM_BACKEND=hip (the HIP backend behind the
reference's LightFM class, evaluation module and tests) or unset (the reference's own extension).

Where the reference's files come from: /root/reference when it exists (this container); else
oracle/_ref/pysuite/, the BYTECODE of exactly those files (`make -C oracle pysuite`: compiled where they
lie, like the C extension next to it; git-ignored, travels to the GPU box with the other built
artefacts).  No reference source is ever written into the repository.
"""
import importlib.machinery
import importlib.util
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
PYSUITE = os.path.join(HERE, "_ref", "pysuite")
PACKAGE_FILES = ("__init__", "lightfm", "evaluation", "data", "cross_validation", "version")
TEST_FILES = ("test_api", "test_evaluation", "test_fast_functions", "test_data", "test_cross_validation",
              "test_movielens")

SHIM = '''\
import os

if os.environ.get("LIGHTFM_BACKEND") == "hip":
    # MI355X backend: same names, same signatures, C ABI of include/lfm_hip.h behind ctypes
    from lightfm_amd._lightfm_fast import *  # noqa: F401,F403
    from lightfm_amd._lightfm_fast import __test_in_positives  # noqa: F401
else:
    try:
        from ._lightfm_fast_openmp import *  # noqa: F401,F403
        from ._lightfm_fast_openmp import __test_in_positives  # noqa: F401
    except ImportError:
        import warnings

        warnings.warn("LightFM was compiled without OpenMP support. Only a single thread will be used.")
        from ._lightfm_fast_no_openmp import *  # noqa: F401,F403
        from ._lightfm_fast_no_openmp import __test_in_positives  # noqa: F401
'''

# A stand-in for lightfm.datasets (the reference downloads MovieLens-100k; there is no network): data of
# the same shape and format -- 943 x 1,682 int32 COO matrices of ratings 1..5, ~90.6 k train / 9.4 k test,
# item feature CSRs (identity and / or 19 genre columns), labels -- drawn from a latent-factor process so
# that the accuracy floors the reference's tests pin are meaningful.  Written by this project.
DATASETS = '''\
import numpy as np
import scipy.sparse as sp

__all__ = ["fetch_movielens", "fetch_stackexchange"]

_N_USERS, _N_ITEMS, _N_GENRES = 943, 1682, 19


def _world():
    """Exposure: a user's rated items are drawn by popularity (a steep power law) x affinity; the rating
    is affinity + an item quality correlated with popularity + a user bias + noise, rounded to 1..5.
    Tuned (oracle/ref_suite.py docstring) so that the reference's CPU extension clears the floors its own
    tests pin on the real MovieLens-100k; exactly 100,000 ratings: 90,570 train + 10 per user test."""
    rng = np.random.RandomState(20260923)
    k = 4
    zu, zi = rng.randn(_N_USERS, k), rng.randn(_N_ITEMS, k)
    genres = (rng.rand(_N_ITEMS, _N_GENRES) < 0.09)
    genres[np.arange(_N_ITEMS), rng.randint(0, _N_GENRES, size=_N_ITEMS)] = True
    zg = rng.randn(_N_GENRES, k)
    zi = 0.6 * zi + 0.8 * (genres @ zg) / np.sqrt(np.maximum(1, genres.sum(axis=1)))[:, None]
    rank = rng.permutation(_N_ITEMS)
    pop = 1.0 / (rank + 20.0) ** 1.8
    quality = 0.85 * (-np.log(rank + 1.0) + np.log(_N_ITEMS) / 2) / 2.0 + np.sqrt(1 - 0.85 ** 2) * rng.randn(_N_ITEMS)
    user_bias = rng.randn(_N_USERS)
    act = np.maximum(20, np.round(rng.lognormal(4.15, 0.95, size=_N_USERS))).astype(np.int64)
    act = np.minimum(act, 737)
    while act.sum() != 100000:
        diff = 100000 - act.sum()
        for i in rng.randint(0, _N_USERS, size=abs(diff)):
            if diff > 0 and act[i] < 737:
                act[i] += 1
            elif diff < 0 and act[i] > 20:
                act[i] -= 1
    rows, cols, vals = [], [], []
    for u in range(_N_USERS):
        aff = zi @ zu[u] / np.sqrt(k)
        p = pop * np.exp(1.8 * aff)
        items = rng.choice(_N_ITEMS, size=act[u], replace=False, p=p / p.sum())
        score = 3.3 + 0.7 * aff[items] + 0.8 * quality[items] + 0.35 * user_bias[u] + 0.95 * rng.randn(len(items))
        rows.append(np.full(len(items), u))
        cols.append(items)
        vals.append(np.clip(np.round(score), 1, 5))
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    # ua.base / ua.test: exactly 10 ratings per user in the test split
    test = np.zeros(len(rows), bool)
    start = 0
    for u in range(_N_USERS):
        n = act[u]
        test[start + rng.choice(n, size=10, replace=False)] = True
        start += n
    return rows, cols, vals, test, genres


def fetch_movielens(data_home=None, indicator_features=True, genre_features=False, min_rating=0.0,
                    download_if_missing=True):
    if not (indicator_features or genre_features):
        raise ValueError("At least one of item_indicator_features or genre_features must be True")
    rows, cols, vals, test, genres = _world()

    def mat(mask):
        keep = mask & (vals >= min_rating)
        return sp.coo_matrix((vals[keep].astype(np.int32), (rows[keep], cols[keep])),
                             shape=(_N_USERS, _N_ITEMS), dtype=np.int32)
    id_features = sp.identity(_N_ITEMS, format="csr", dtype=np.float32)
    genre_mat = sp.csr_matrix(genres.astype(np.float32))
    item_labels = np.array(["item %d" % i for i in range(_N_ITEMS)], dtype=object)
    genre_labels = np.array(["genre:%d" % g for g in range(_N_GENRES)], dtype=object)
    if indicator_features and not genre_features:
        features, labels = id_features, item_labels
    elif genre_features and not indicator_features:
        features, labels = genre_mat, genre_labels
    else:
        features = sp.hstack([id_features, genre_mat]).tocsr().astype(np.float32)
        labels = np.concatenate((item_labels, genre_labels))
    return {"train": mat(~test), "test": mat(test), "item_features": features,
            "item_feature_labels": labels, "item_labels": item_labels}


def fetch_stackexchange(*args, **kwargs):
    raise IOError("no network: fetch_stackexchange is not available in this harness")
'''


def build_pysuite():
    """Bytecode of the reference's host layer and tests -> oracle/_ref/pysuite (needs /root/reference)."""
    subprocess.check_call(["make", "-C", HERE, "pysuite"], stdout=subprocess.DEVNULL)


def available():
    return os.path.isdir(os.path.join(REF, "lightfm")) or os.path.exists(os.path.join(PYSUITE, "lightfm", "lightfm.pyc"))


_WRAPPER = '''\
# collects the reference's test module from its bytecode (no source travels to the GPU box)
import importlib.util as _u
_spec = _u.spec_from_file_location("_ref_%(name)s", %(path)r)
_mod = _u.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
globals().update({k: v for k, v in vars(_mod).items() if not k.startswith("__")})
'''


def materialise(dst):
    """Lays the scratch package + tests out under `dst` (a directory outside the repository)."""
    assert not os.path.abspath(dst).startswith(ROOT + os.sep), "the scratch copy must not live in the repository"
    pkg, tests = os.path.join(dst, "lightfm"), os.path.join(dst, "tests")
    os.makedirs(os.path.join(pkg, "datasets"), exist_ok=True)
    os.makedirs(tests, exist_ok=True)
    from_source = os.path.isdir(os.path.join(REF, "lightfm"))
    for name in PACKAGE_FILES:
        if from_source:
            shutil.copy(os.path.join(REF, "lightfm", name + ".py"), os.path.join(pkg, name + ".py"))
        else:  # sourceless import: <name>.pyc next to where <name>.py would be
            shutil.copy(os.path.join(PYSUITE, "lightfm", name + ".pyc"), os.path.join(pkg, name + ".pyc"))
    for name in TEST_FILES:
        if from_source:
            shutil.copy(os.path.join(REF, "tests", name + ".py"), os.path.join(tests, name + ".py"))
        else:
            pyc = os.path.join(tests, "_ref_" + name + ".pyc")
            shutil.copy(os.path.join(PYSUITE, "tests", name + ".pyc"), pyc)
            with open(os.path.join(tests, name + ".py"), "w") as f:
                f.write(_WRAPPER % {"name": name, "path": pyc})
    open(os.path.join(tests, "__init__.py"), "w").close()
    with open(os.path.join(pkg, "_lightfm_fast.py"), "w") as f:
        f.write(SHIM)
    with open(os.path.join(pkg, "datasets", "__init__.py"), "w") as f:
        f.write(DATASETS)
    so = os.path.join(HERE, "_ref", "strict", "_lightfm_fast_openmp.so")
    if os.path.exists(so):
        # the extension module must be importable as lightfm._lightfm_fast_openmp
        suffix = importlib.machinery.EXTENSION_SUFFIXES[0]
        shutil.copy(so, os.path.join(pkg, "_lightfm_fast_openmp" + suffix))
    return dst


def run(dst, backend, files=TEST_FILES, mode=None, extra_env=None, timeout=3000, select=None):
    """pytest over the scratch suite in a subprocess.  backend "hip" | "reference".  Returns
    ({test id: outcome}, pytest's exit code, the tail of its output)."""
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([dst, ROOT] + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else []))
    env.pop("LIGHTFM_BACKEND", None)
    if backend == "hip":
        env["LIGHTFM_BACKEND"] = "hip"
    if mode:
        env["LIGHTFM_AMD_MODE"] = mode
    env.update(extra_env or {})
    report = os.path.join(dst, "report_%s.txt" % backend)
    cmd = [sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "-rA", "--tb=short",
           "-o", "python_files=test_*.py", "--rootdir", dst] + [os.path.join(dst, "tests", f + ".py") for f in files]
    if select:
        cmd += ["-k", select]
    p = subprocess.run(cmd, cwd=dst, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    text = p.stdout.decode("utf-8", "replace")
    with open(report, "w") as f:
        f.write(text)
    outcomes = {}
    for line in text.splitlines():
        parts = line.split()
        if len(parts) >= 2 and parts[0] in ("PASSED", "FAILED", "ERROR", "SKIPPED", "XFAIL", "XPASS") and "::" in parts[1]:
            outcomes[parts[1].replace(dst + os.sep, "")] = parts[0]
    return outcomes, p.returncode, text[-6000:]


def loaded_backend(dst, backend):
    """Which native module the scratch package's `lightfm._lightfm_fast` resolves to (sanity of the shim)."""
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([dst, ROOT])
    env.pop("LIGHTFM_BACKEND", None)
    if backend == "hip":
        env["LIGHTFM_BACKEND"] = "hip"
    out = subprocess.check_output([sys.executable, "-c",
                                   "import lightfm._lightfm_fast as f, lightfm.lightfm as l, json;"
                                   "print(json.dumps([f.fit_warp.__module__, l.fit_warp.__module__, l.__file__]))"],
                                  env=env, cwd=dst)
    return json.loads(out.decode().strip().splitlines()[-1])
