"""End-to-end LightFM.fit wall time at the ML-20M shape (host prologue + uploads + epochs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lightfm_amd import LightFM, options, synthetic
data = synthetic.named("ml-20m")
for dev in (True, False):
    options.set(device_shuffle=dev)
    m = LightFM(no_components=64, loss="warp", random_state=1)
    t = time.time(); m.fit(data, epochs=1); t1 = time.time() - t
    t = time.time(); m.fit_partial(data, epochs=10); t10 = time.time() - t
    k = sum(s["kernel_ms"] for s in m._last_epoch_stats) / 1e3
    print("device_shuffle=%s: fit(1 epoch) %.2fs; fit_partial(10 epochs) %.2fs = %.1f M interactions/s end to end (kernels %.2fs)"
          % (dev, t1, t10, data.nnz * 10 / t10 / 1e6, k), flush=True)
