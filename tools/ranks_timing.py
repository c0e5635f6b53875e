"""Wall and device time of predict_rank / precision_at_k at the ML-20M shape.

    python tools/ranks_timing.py [n_users (default: all 138,493)]     env RANKS_TIMING_MODES=3,2: kernels to time (LIGHTFM_AMD_RANKS_MFMA values), RANKS_TIMING_D: no_components
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from lightfm_amd import LightFM, synthetic, _native as N
from lightfm_amd.evaluation import precision_at_k
data = synthetic.named("ml-20m")
n_eval = int(sys.argv[1]) if len(sys.argv) > 1 else data.shape[0]
train, test = synthetic.train_test_split(data, 0.1, seed=1)
users = np.sort(np.random.RandomState(0).choice(data.shape[0], size=n_eval, replace=False))
mask = np.zeros(data.shape[0], bool); mask[users] = True
keep = mask[test.row]
test_sub = sp.coo_matrix((test.data[keep], (test.row[keep], test.col[keep])), shape=test.shape, dtype=np.float32).tocsr()
D = int(os.environ.get("RANKS_TIMING_D", "64"))
m = LightFM(no_components=D, loss="warp", random_state=1).fit(train, epochs=2)
train_csr = train.tocsr()
modes = os.environ.get("RANKS_TIMING_MODES", "3").split(",")   # LIGHTFM_AMD_RANKS_MFMA values, e.g. "3,2"
seen = {}
for mode in modes:
    os.environ["LIGHTFM_AMD_RANKS_MFMA"] = mode
    for _ in range(3):
        t = time.time(); pk = precision_at_k(m, test_sub, train_interactions=train_csr, k=10, check_intersections=False); dt = time.time() - t
        p = pk.mean()
        kms = N.lib().lfm_last_kernel_ms()
        pairs = float(n_eval) * data.shape[1]
        print("d %d mode %s: precision_at_k over %d users x %d items (%d test interactions): wall %.2fs, kernels %.1f ms = %.1f G user-item "
              "scores/s (%.2f TFLOP/s of 2*d flops, %.3f of the 157.3 TFLOP/s fp32 matrix peak), p@10 %.4f" % (D, mode, n_eval, data.shape[1],
              test_sub.nnz, dt, kms, pairs / kms / 1e6, 2 * D * pairs / kms / 1e9, 2 * D * pairs / kms / 1e9 / 157.3, p), flush=True)
    seen[mode] = pk
if len(seen) > 1:
    print("identical precision vectors across modes:", all(np.array_equal(seen[modes[0]], v) for v in seen.values()))
