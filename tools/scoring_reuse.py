"""Wall time of repeated predict_rank / predict calls at the ML-20M shape with the scoring session kept on the
model between calls (default) and without (options.cache_scoring_session = False: tables uploaded per call).

    python tools/scoring_reuse.py [calls (default 20)]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lightfm_amd import LightFM, synthetic, options, _native as N
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20
data = synthetic.named("ml-20m")
train, test = synthetic.train_test_split(data, 0.1, seed=1)
m = LightFM(no_components=64, loss="warp", random_state=1).fit(train, epochs=1)
test_csr, train_csr = test.tocsr(), train.tocsr()
uids = np.random.RandomState(0).randint(0, data.shape[0], size=1_000_000).astype(np.int32)
iids = np.random.RandomState(1).randint(0, data.shape[1], size=1_000_000).astype(np.int32)
for cache in (True, False, True):
    options.set(cache_scoring_session=cache)
    m._drop_scoring_session()
    m.predict_rank(test_csr, train_interactions=train_csr, check_intersections=False)  # warm-up (and the first upload)
    t = time.time(); kms = 0.0
    for _ in range(calls):
        m.predict_rank(test_csr, train_interactions=train_csr, check_intersections=False)
        kms += N.lib().lfm_last_kernel_ms()
    dt = time.time() - t
    t = time.time()
    for _ in range(calls):
        m.predict(uids, iids)
    dp = time.time() - t
    sig = time.time(); [m._array_signature(getattr(m, n)) for n in m._SCORED]; sig = time.time() - sig
    print("cache_scoring_session=%s: predict_rank over %d x %d (%d test interactions): %.1f ms per call wall, %.1f ms of "
          "it kernels; predict of 1 M pairs: %.1f ms per call; the four-array checksum %.1f ms"
          % (cache, data.shape[0], data.shape[1], test_csr.nnz, dt * 1e3 / calls, kms / calls, dp * 1e3 / calls, sig * 1e3), flush=True)
