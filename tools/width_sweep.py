"""Epoch rates of WARP / BPR / logistic identity models over no_components at the ML-20M shape (kernel time per epoch, epochs 3..5).
    python tools/width_sweep.py [losses=warp,bpr,logistic] [widths=10,16,20,32,48,64,100,128,200,256]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from lightfm_amd import LightFM, synthetic
losses = (sys.argv[1] if len(sys.argv) > 1 else "warp,bpr,logistic").split(",")
widths = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "10,16,20,32,48,64,100,128,200,256").split(",")]
data = synthetic.named("ml-20m")
rng = np.random.RandomState(3)
signed = sp.coo_matrix((np.where(rng.rand(data.nnz) < 0.5, 1.0, -1.0).astype(np.float32), (data.row, data.col)), shape=data.shape, dtype=np.float32)
for loss in losses:
    for d in widths:
        mat = signed if loss == "logistic" else data
        m = LightFM(no_components=d, loss=loss, random_state=1)
        m.fit_partial(mat, epochs=2)
        m.fit_partial(mat, epochs=3)
        st = m._last_epoch_stats
        ms = np.mean([s["kernel_ms"] for s in st])
        print("%-9s d=%-3d kernel %7.2f ms/epoch  %7.1f M interactions/s  (kernel_used %d, plan flags %d, tile_ng %d, in flight %d, user rows by plain stores %d)"
              % (loss, d, ms, mat.nnz / ms / 1e3, st[-1]["kernel_used"], st[-1]["plan_flags"], st[-1]["tile_ng"], st[-1]["in_flight"], st[-1]["user_store"]), flush=True)
