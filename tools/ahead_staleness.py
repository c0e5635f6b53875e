"""the steady-state tile kernel (warp_tile_ahead.hpp) requests the rows of pass t + 1 before pass t publishes,
so consecutive positions of one wavefront that share a user or an item always score on pre-update rows -- one pass of
systematic staleness the plain tile kernel (options.debug bit 10 = 1024) does not have.  Bounded here where it would
show: small, high-collision problems (few users and items, d = 64, WARP), precision@10 on a held-out tenth and train
AUC, N seeds per arm, plus the reference (16 threads) as the yardstick.

    python tools/ahead_staleness.py [seeds=12] [epochs=10]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from lightfm_amd import LightFM, options, synthetic
from lightfm_amd.evaluation import auc_score, precision_at_k
from oracle.ref_model import RefLightFM

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for nu, ni, nnz in ((300, 120, 12000), (1000, 400, 60000), (4000, 300, 200000)):
    data = synthetic.make_interactions(nu, ni, nnz)
    train, test = synthetic.train_test_split(data, 0.1, seed=1)
    rows = {}
    for arm, cls, debug in (("ahead (default)", LightFM, 4), ("plain tile kernel", LightFM, 4 | 1024), ("reference, 16 threads", RefLightFM, 0)):
        res = []
        for seed in range(1, n_seeds + 1):
            # AHEAD_RAMP=1: the shipped concurrency ramp; default here: the whole chip's width from the first interaction (worst case)
            options.set(mode="parallel", debug=debug, ramp_k=0 if os.environ.get("AHEAD_RAMP") else -1)
            m = cls(no_components=64, loss="warp", random_state=seed)
            m.fit(train, epochs=epochs, num_threads=16 if cls is RefLightFM else 1)
            if cls is LightFM:
                assert m._last_epoch_stats[-1]["kernel_used"] == 1
            res.append((precision_at_k(m, test, train_interactions=train, k=10).mean(), auc_score(m, train).mean()))
        options.set(debug=0, ramp_k=0)
        r = np.array(res)
        rows[arm] = r
        print("%d x %d x %d  %-24s p@10 test %.4f (sem %.4f)  train AUC %.4f (sem %.4f)  n=%d" % (
            nu, ni, train.nnz, arm, r[:, 0].mean(), r[:, 0].std() / np.sqrt(len(r)), r[:, 1].mean(),
            r[:, 1].std() / np.sqrt(len(r)), len(r)), flush=True)
    d = rows["ahead (default)"].mean(axis=0) - rows["plain tile kernel"].mean(axis=0)
    print("%d x %d: ahead - plain: p@10 %+.4f, train AUC %+.4f" % (nu, ni, d[0], d[1]), flush=True)
