"""predict_rank of the default kernel against the scalar sequential-dot kernel on many small random problems with
dense test matrices (every item a test item: 4+ passes per user, scores close together); prints every mismatch.

    python tools/ranks_stress.py [cases (default 200)]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from lightfm_amd import LightFM

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(7)
bad = bad4 = 0
for c in range(cases):
    nu, ni, d = int(rng.randint(5, 70)), int(rng.randint(20, 400)), int(rng.choice([3, 10, 32, 33, 64, 100]))
    train = sp.rand(nu, ni, density=0.1, format="csr", random_state=int(rng.randint(1 << 30)))
    train.data[:] = 1.0
    m = LightFM(no_components=d, random_state=int(rng.randint(1 << 30))).fit_partial(train.tocoo(), epochs=int(rng.randint(1, 4)))
    if c % 3 == 0:  # tight scores: everything inside the rounding band
        m.item_embeddings *= 1e-3
    dense = sp.csr_matrix(np.ones((nu, ni), np.float32))
    if c % 5 == 4:  # a sparse test matrix that INTERSECTS the train matrix (train positives among the test items)
        dense = (sp.rand(nu, ni, density=0.3, format="csr", random_state=c) + train).tocsr()
        dense.data[:] = 1.0
    tr = train if c % 2 else None
    out = {}
    for mode in ("4", "3", "0"):
        os.environ["LIGHTFM_AMD_RANKS_MFMA"] = mode
        out[mode] = m.predict_rank(dense, train_interactions=tr, check_intersections=False).toarray()
    if not np.array_equal(out["4"], out["0"]):
        bad4 += 1
        u, i = np.nonzero(out["4"] != out["0"])
        print("case %d (nu %d ni %d d %d train %s): %d cells differ between the bf16-pipe sweep and the scalar kernel; first: user %d item %d: %.0f vs %.0f"
              % (c, nu, ni, d, tr is not None, len(u), u[0], i[0], out["4"][u[0], i[0]], out["0"][u[0], i[0]]), flush=True)
    if not np.array_equal(out["3"], out["0"]):
        bad += 1
        u, i = np.nonzero(out["3"] != out["0"])
        print("case %d (nu %d ni %d d %d train %s): %d cells differ; first: user %d item %d bucket-search %.0f scalar %.0f"
              % (c, nu, ni, d, tr is not None, len(u), u[0], i[0], out["3"][u[0], i[0]], out["0"][u[0], i[0]]), flush=True)
        s = m.predict(np.repeat(u[0], ni), np.arange(ni))
        order = np.argsort(-s)
        pos = int(np.where(order == i[0])[0][0])
        print("   scores around it:", [(int(order[k]), float(s[order[k]])) for k in range(max(0, pos - 2), min(ni, pos + 3))], flush=True)
os.environ.pop("LIGHTFM_AMD_RANKS_MFMA", None)
print("%d of %d cases differ (fp32 products), %d (bf16 pipe)" % (bad, cases, bad4))

# the shape of tests/test_lightfm_api.py::test_predict_rank_known_answers, many unseeded models: both kernels, and
# whether the ranks of a row are a permutation (an exact tie of two float32 scores gives a repeated rank in the
# reference too)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nu, ni = 10, 100
train = sp.rand(nu, ni, density=0.1, format="csr", random_state=42)
train.data[:] = 1.0
dense = sp.csr_matrix(np.ones((nu, ni), np.float32))
differ = differ4 = ties = 0
for c in range(reps):
    m = LightFM().fit_partial(train)
    out = {}
    for mode in ("4", "3", "0"):
        os.environ["LIGHTFM_AMD_RANKS_MFMA"] = mode
        out[mode] = m.predict_rank(dense, num_threads=2).toarray()
    if not np.array_equal(out["4"], out["0"]):
        differ4 += 1
    if not np.array_equal(out["3"], out["0"]):
        differ += 1
        u, i = np.nonzero(out["3"] != out["0"])
        print("api case %d: %d cells differ; first: user %d item %d bucket-search %.0f scalar %.0f" % (c, len(u), u[0], i[0], out["3"][u[0], i[0]], out["0"][u[0], i[0]]), flush=True)
    for row in range(nu):
        if not np.array_equal(np.sort(out["0"][row]), np.arange(ni)):
            ties += 1
            s = m.predict(np.repeat(row, ni), np.arange(ni))
            print("api case %d row %d: scalar ranks not a permutation; distinct scores %d of %d" % (c, row, len(np.unique(s)), ni), flush=True)
os.environ.pop("LIGHTFM_AMD_RANKS_MFMA", None)
if reps:
    print("api shape: %d (fp32 products) / %d (bf16 pipe) of %d models differ between the kernels; %d rows with repeated ranks in the scalar kernel" % (differ, differ4, reps, ties))
