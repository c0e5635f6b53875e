"""Quick timing of the generic epoch kernels (parallel mode): BPR / logistic / k-OS, identity and
tag features, on an ML-20M-shaped 4M-interaction sample."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from lightfm_amd import LightFM, options, synthetic
data = synthetic.named("ml-20m", scale=0.2)
tags = synthetic.tag_item_features(data.shape[1])
for loss, d, feats in (("bpr", 128, None), ("bpr", 128, tags), ("logistic", 64, None), ("warp-kos", 64, None), ("warp-kos", 128, tags), ("warp", 64, tags)):
    m = LightFM(no_components=d, loss=loss, random_state=1)
    m.fit_partial(data, item_features=feats, epochs=1)
    m.fit_partial(data, item_features=feats, epochs=2)
    ms = np.mean([s["kernel_ms"] for s in m._last_epoch_stats])
    print("%-9s d=%-3d feats=%-5s kernel %.2f ms/epoch  %.1f M interactions/s" % (loss, d, "tags" if feats is not None else "id", ms, data.nnz / ms / 1e3), flush=True)
