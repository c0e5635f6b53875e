"""precision@10 at the reference's default width (no_components = 10) on the C2-regime gate data: the reference (16 threads), this
backend with the narrow-model tile kernel (default) and with the wide one (LIGHTFM_AMD_TILE_PAIRS=0 is read per process: the wide arm
runs with options.debug bit 10 = the plain tile kernel AND in a second process through the env).   python tools/narrow_quality.py [seeds=32]"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from lightfm_amd import LightFM, synthetic
from lightfm_amd.evaluation import precision_at_k
from oracle.ref_model import RefLightFM

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
data = synthetic.make_interactions(17312, 13372, 2_500_000, seed=11)
train, test = synthetic.train_test_split(data, 0.1, seed=1)
tr, te = train.tocsr(), test.tocsr()
p10 = lambda m: float(precision_at_k(m, te, train_interactions=tr, k=10).mean())
res = {}
if os.environ.get("NARROW_QUALITY_REF", "1") != "0":
    def fit_ref(seed):
        r = RefLightFM(no_components=10, loss="warp", random_state=seed)
        r.fit(train, epochs=5, num_threads=16)
        return p10(r)
    with ThreadPoolExecutor(max_workers=3) as pool:
        res["reference"] = list(pool.map(fit_ref, range(1, n + 1)))
out = []
for seed in range(1, n + 1):
    m = LightFM(no_components=10, loss="warp", random_state=seed)
    m.fit(train, epochs=5)
    out.append(p10(m))
res["hip (TILE_PAIRS=%s, ROW_PAIRS=%s, flags %s)" % (os.environ.get("LIGHTFM_AMD_TILE_PAIRS", "1"), os.environ.get("LIGHTFM_AMD_ROW_PAIRS", "2"), m._last_epoch_stats[-1].get("plan_flags"))] = out
for k, v in res.items():
    print("%-34s %.4f +- %.4f (s.e., n=%d)" % (k, np.mean(v), np.std(v, ddof=1) / np.sqrt(len(v)), len(v)), flush=True)
