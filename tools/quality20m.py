"""precision@10 parity at the bench workload's shape (ML-20M-shaped synthetic data):
reference Cython/OpenMP path vs the HIP backend (update modes / kernels).  GPU box.

    python tools/quality20m.py [epochs] [n_eval_users] [seeds] [variants]

precision@10 is evaluated with the reference's metric (precision_at_k over predict_rank)
on a fixed random subset of users (the O(users x items) rank pass is the expensive part).
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

from lightfm_amd import LightFM, options, synthetic
from lightfm_amd.evaluation import precision_at_k
from oracle.ref_model import RefLightFM

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_eval = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
seeds = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1]
variants = sys.argv[4].split(",") if len(sys.argv) > 4 else ["ref", "hip0", "hip1"]
scale = float(os.environ.get("QUALITY_SCALE", "1.0"))

t0 = time.time()
data = synthetic.named("ml-20m", scale=scale)
train, test = synthetic.train_test_split(data, 0.1, seed=1)
users = np.sort(np.random.RandomState(0).choice(data.shape[0], size=n_eval, replace=False))
mask = np.zeros(data.shape[0], bool)
mask[users] = True


def rows_subset(coo):
    keep = mask[coo.row]
    return sp.coo_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=coo.shape,
                         dtype=np.float32).tocsr()


test_sub, train_sub = rows_subset(test), rows_subset(train)
print("data %s train %d test %d eval users %d (%.0fs)" % (data.shape, train.nnz, test.nnz, n_eval,
                                                          time.time() - t0), flush=True)


def evaluate(m):
    pte = precision_at_k(m, test_sub, train_interactions=train, k=10).mean()
    ptr = precision_at_k(m, train_sub, k=10).mean()
    return ptr, pte


for v in variants:
    res = []
    for seed in seeds:
        t = time.time()
        if v == "ref":
            m = RefLightFM(no_components=64, loss="warp", random_state=seed)
            m.fit(train, epochs=epochs, num_threads=min(16, os.cpu_count()))
        elif v == "ref1":
            m = RefLightFM(no_components=64, loss="warp", random_state=seed)
            m.fit(train, epochs=epochs, num_threads=1)
        else:
            # hip<update_mode>[g] : g = generic kernel
            options.set(mode="parallel", update_mode=int(v[3]), warp_kernel=1 if v.endswith("g") else 0,
                        debug=int(os.environ.get("QUALITY_DEBUG", "0")))  # QUALITY_DEBUG: lfm_opts.debug bits (kernel experiments)
            m = LightFM(no_components=64, loss="warp", random_state=seed)
            m.fit(train, epochs=epochs, num_threads=1)
        dt = time.time() - t
        ptr, pte = evaluate(m)
        res.append((ptr, pte, dt))
        print("  %s seed %d: p@10 train %.4f test %.4f  fit %.1fs" % (v, seed, ptr, pte, dt), flush=True)
    r = np.array(res)
    print("%-6s p@10 train %.4f test %.4f (std %.4f) fit %.1fs" % (v, r[:, 0].mean(), r[:, 1].mean(),
                                                                    r[:, 1].std(), r[:, 2].mean()),
          flush=True)
