"""precision@10 at the FULL C3 shape (ML-20M shape, BPR d = 128, [identity | 8 tags of 1 128]): this backend with its defaults
(3 seeds) against the reference's compiled OpenMP build, 16 threads (N_REF seeds; ~150 s of CPU each), 3 epochs, all users with test
interactions.  bench.py's own quality leg for c3 runs a 1/8 row sub-sample (where the plain-store user rows are off by rule).

    python tools/quality_c3_full.py [n_ref_seeds=1]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from lightfm_amd import LightFM, synthetic
from lightfm_amd.evaluation import precision_at_k
from oracle.ref_model import RefLightFM

n_ref = int(sys.argv[1]) if len(sys.argv) > 1 else 1
data = synthetic.make_interactions(138493, 26744, 21000000, seed=42)
train, test = synthetic.split_off_test(data, 20000263, seed=1)
feats = synthetic.tag_item_features(26744)
tr, te = train.tocsr(), test.tocsr()


def p10(m):
    return float(precision_at_k(m, te, train_interactions=tr, k=10, item_features=feats).mean())


hip = []
for seed in (7, 8, 9):
    m = LightFM(no_components=128, loss="bpr", random_state=seed)
    m.fit(train, item_features=feats, epochs=3)
    hip.append(p10(m))
    print("hip seed %d: %.5f (user rows stored: %s)" % (seed, hip[-1], m._last_epoch_stats[-1].get("user_store")), flush=True)
ref = []
for seed in range(7, 7 + n_ref):
    t = time.time()
    r = RefLightFM(no_components=128, loss="bpr", random_state=seed)
    r.fit(train, item_features=feats, epochs=3, num_threads=min(16, os.cpu_count() or 1))
    ref.append(p10(r))
    print("ref seed %d: %.5f (%.0f s)" % (seed, ref[-1], time.time() - t), flush=True)
if ref:
    print("C3 full size: hip %.5f (n=3) ref %.5f (n=%d) delta %+.5f" % (np.mean(hip), np.mean(ref), len(ref), np.mean(hip) - np.mean(ref)))
else:
    print("C3 full size: hip %.5f (n=3; std %.5f)  plan_flags %s  [%s]" % (np.mean(hip), np.std(hip), m._last_epoch_stats[-1].get("plan_flags"), " ".join("%s=%s" % kv for kv in sorted(os.environ.items()) if kv[0].startswith("LIGHTFM_AMD_"))))
