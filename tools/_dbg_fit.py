import numpy as np, scipy.sparse as sp, time, sys, os
sys.path.insert(0, ".")
os.environ["LIGHTFM_AMD_TIMING"] = "1"
from lightfm_amd import LightFM, synthetic
data = synthetic.named("ml-20m")
rng = np.random.RandomState(3)
signed = sp.coo_matrix((np.where(rng.rand(data.nnz) < 0.5, 1.0, -1.0).astype(np.float32), (data.row, data.col)), shape=data.shape, dtype=np.float32)
for rep in range(3):
    m = LightFM(random_state=1)
    t = time.perf_counter(); m.fit(signed, epochs=10); dt = time.perf_counter() - t
    print("default model fit(10 epochs) %.1f ms = %.2f G interactions/s; kernels %s" % (dt * 1e3, 10 * data.nnz / dt / 1e9, [round(s["kernel_ms"], 1) for s in m._last_epoch_stats]), flush=True)
for rep in range(2):
    m = LightFM(no_components=64, loss="warp", random_state=1)
    t = time.perf_counter(); m.fit(data, epochs=10); dt = time.perf_counter() - t
    print("c2 fit(10 epochs) %.1f ms = %.2f G interactions/s; kernels %s" % (dt * 1e3, 10 * data.nnz / dt / 1e9, [round(s["kernel_ms"], 1) for s in m._last_epoch_stats]), flush=True)
