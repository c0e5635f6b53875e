"""Experiment: does a low-concurrency first epoch (accumulators still ~1) let later epochs run
with many interactions in flight?  ML-100k shape, optional tag features."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lightfm_amd import LightFM, options, synthetic
from lightfm_amd.evaluation import precision_at_k
tags = os.environ.get("QUALITY_TAGS")
data = synthetic.named("ml-100k")
train, test = synthetic.train_test_split(data, 0.1, seed=1)
itf = None
if tags:
    a, b = [int(x) for x in tags.split(",")]
    itf = synthetic.tag_item_features(data.shape[1], n_tags=a, per_item=b)
def run(warm_cap, warm_epochs, cap, epochs=10):
    res = []
    for seed in (1, 2, 3):
        m = LightFM(no_components=32, loss="warp", random_state=seed)
        if warm_epochs:
            options.set(max_waves=warm_cap)
            m.fit_partial(train, item_features=itf, epochs=warm_epochs)
        options.set(max_waves=cap)
        m.fit_partial(train, item_features=itf, epochs=epochs - warm_epochs)
        res.append(precision_at_k(m, test, train_interactions=train, k=10, item_features=itf).mean())
    print("warm %d epochs at %4d then cap %5d: p@10 test %.4f (std %.4f)" % (warm_epochs, warm_cap, cap, np.mean(res), np.std(res)), flush=True)
for cap in (16, 117, 1024, 8192) if os.environ.get('WARMUP_SWEEP') else (0,):
    run(0, 0, cap)
    if os.environ.get('WARMUP_SWEEP'):
        run(8, 1, cap)
        run(8, 2, cap)


