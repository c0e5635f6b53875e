"""precision@10 of the hybrid gate problems (tests/test_precision_parity.py) against the hot set's record length: the
reference (16 threads) once per problem, then this backend per arm (LIGHTFM_AMD_HOT_* are read per epoch).

    python tools/hot_gate_sweep.py [seeds=8] ARM [ARM ...]      ARM = name:ENV=VALUE,ENV=VALUE   (name "off": HOT_SLICES=0)
"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from lightfm_amd import LightFM, options, synthetic
from lightfm_amd.evaluation import precision_at_k
from oracle.ref_model import RefLightFM

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
arms = [a for a in sys.argv[1:] if ":" in a]
PROBLEMS = [
    ("bpr-1128tags-d128", "bpr", 128, (8656, 13372, 1_000_000), dict(n_tags=1128, per_item=8), 3, {}),
    ("warp-200tags-d64", "warp", 64, (8656, 6686, 1_000_000), dict(n_tags=200, per_item=4), 5, {}),
    ("kos-100tags-d64", "warp-kos", 64, (4000, 3000, 300_000), dict(n_tags=100, per_item=4), 5, {}),
]
only = os.environ.get("SWEEP_PROBLEMS")
for name, loss, d, (nu, ni, nnz), tagkw, epochs, kw in PROBLEMS:
    if only and name not in only.split(","):
        continue
    data = synthetic.make_interactions(nu, ni, nnz, seed=11)
    train, test = synthetic.train_test_split(data, 0.1, seed=1)
    feats = synthetic.tag_item_features(ni, **tagkw)
    tr, te = train.tocsr(), test.tocsr()

    def p10(m):
        return float(precision_at_k(m, te, train_interactions=tr, k=10, item_features=feats).mean())

    def fit_ref(seed):
        r = RefLightFM(no_components=d, loss=loss, random_state=seed)
        r.fit(train, item_features=feats, epochs=epochs, num_threads=16)
        return r

    with ThreadPoolExecutor(max_workers=3) as pool:
        pending = [pool.submit(fit_ref, seed) for seed in range(1, n_seeds + 1)]
        res = {}
        for arm in arms:
            aname, envs = arm.split(":", 1)
            for k in [k for k in os.environ if k.startswith("LIGHTFM_AMD_HOT_")]:
                del os.environ[k]
            for kv in envs.split(","):
                if "=" in kv:
                    k, v = kv.split("=")
                    os.environ["LIGHTFM_AMD_" + k] = v
            out = []
            for seed in range(1, n_seeds + 1):
                m = LightFM(no_components=d, loss=loss, random_state=seed)
                m.fit(train, item_features=feats, epochs=epochs)
                out.append(p10(m))
            res[aname] = (np.mean(out), np.std(out, ddof=1) / np.sqrt(len(out)), m._last_epoch_stats[-1].get("plan_flags"), m._last_epoch_stats[-1].get("launches"))
        ref = [p10(f.result()) for f in pending]
    rm, rs = np.mean(ref), np.std(ref, ddof=1) / np.sqrt(len(ref))
    print("%s: reference %.4f +- %.4f (n=%d)" % (name, rm, rs, len(ref)), flush=True)
    for aname, (m_, s_, fl, ln) in res.items():
        print("   %-28s %.4f +- %.4f  delta %+.4f  (flags %s, launches of the last epoch %s)" % (aname, m_, s_, m_ - rm, fl, ln), flush=True)
