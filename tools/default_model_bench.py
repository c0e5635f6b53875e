"""Epoch rates of the identity-feature models the tile kernels do not cover, at the ML-20M shape: the reference's literal default
(LightFM(): logistic, no_components = 10) and BPR, at d = 10 and 64.   python tools/default_model_bench.py [scale=1.0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from lightfm_amd import LightFM, synthetic
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
data = synthetic.named("ml-20m", scale=scale)
rng = np.random.RandomState(3)
signed = sp.coo_matrix((np.where(rng.rand(data.nnz) < 0.5, 1.0, -1.0).astype(np.float32), (data.row, data.col)), shape=data.shape, dtype=np.float32)
for loss, d, mat in (("logistic", 10, signed), ("logistic", 32, signed), ("logistic", 64, signed), ("logistic", 128, signed), ("bpr", 10, data), ("bpr", 32, data), ("bpr", 64, data), ("bpr", 128, data), ("warp", 10, data)):
    m = LightFM(no_components=d, loss=loss, random_state=1)
    m.fit_partial(mat, epochs=2)
    m.fit_partial(mat, epochs=3)
    st = m._last_epoch_stats
    ms = np.mean([s["kernel_ms"] for s in st])
    print("%-9s d=%-3d kernel %.2f ms/epoch  %.1f M interactions/s  (kernel_used %d, plan flags %d, in flight %d, user rows by plain stores %d)"
          % (loss, d, ms, mat.nnz / ms / 1e3, st[-1]["kernel_used"], st[-1]["plan_flags"], st[-1]["in_flight"], st[-1]["user_store"]), flush=True)
