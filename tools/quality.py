"""precision@10 of the HIP backend vs the reference CPU path on the same synthetic
data, across in-flight-wave caps (staleness study).  Runs on the GPU box.

    python tools/quality.py [dataset] [epochs] [d]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from lightfm_amd import LightFM, options, synthetic
from lightfm_amd.evaluation import precision_at_k
from oracle.ref_model import RefLightFM

name = sys.argv[1] if len(sys.argv) > 1 else "ml-100k"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d = int(sys.argv[3]) if len(sys.argv) > 3 else 32
loss = sys.argv[4] if len(sys.argv) > 4 else "warp"
caps = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0, 64, 256, 1024, 8192]
SEEDS = [int(x) for x in os.environ.get("QUALITY_SEEDS", "1,2,3").split(",")]
REF_THREADS = [int(x) for x in os.environ.get("QUALITY_REF_THREADS", "1,16").split(",")]

if name in synthetic.SHAPES:
    data = synthetic.named(name)
else:
    nu, ni, nnz = [int(x) for x in name.split("x")]
    data = synthetic.make_interactions(nu, ni, nnz)
train, test = synthetic.train_test_split(data, 0.1, seed=1)
item_features = None
if os.environ.get("QUALITY_TAGS"):  # hybrid model: [identity | tags] item features
    n_tags, per_item = [int(x) for x in os.environ["QUALITY_TAGS"].split(",")]
    item_features = synthetic.tag_item_features(data.shape[1], n_tags=n_tags, per_item=per_item)
print("data", name, data.shape, "train", train.nnz, "test", test.nnz, flush=True)


def evaluate(m):
    ptr = precision_at_k(m, train, k=10, item_features=item_features).mean()
    pte = precision_at_k(m, test, train_interactions=train, k=10, item_features=item_features).mean()
    return ptr, pte


for threads in [min(t, os.cpu_count()) for t in REF_THREADS]:
    res = []
    for seed in SEEDS:
        m = RefLightFM(no_components=d, loss=loss, random_state=seed)
        t = time.time()
        m.fit(train, item_features=item_features, epochs=epochs, num_threads=threads)
        dt = time.time() - t
        res.append(evaluate(m) + (dt,))
    r = np.array(res)
    print("ref threads=%-3d p@10 train %.4f test %.4f (std %.4f, sem %.4f, n=%d)  %.2fs/fit  %.3g inter/s" % (
        threads, r[:, 0].mean(), r[:, 1].mean(), r[:, 1].std(), r[:, 1].std() / np.sqrt(len(r)), len(r), r[:, 2].mean(),
        train.nnz * epochs / r[:, 2].mean()), flush=True)

modes = [int(x) for x in os.environ.get("QUALITY_MODES", "1,3").split(",")]
for cap, um in [(c, u) for c in caps for u in modes]:
    res = []
    for seed in SEEDS:
        options.set(mode="parallel", max_waves=cap, update_mode=um)
        m = LightFM(no_components=d, loss=loss, random_state=seed)
        m.fit(train, item_features=item_features, epochs=epochs, num_threads=1)
        ms = sum(s["kernel_ms"] for s in m._last_epoch_stats)
        draws = sum(s["counters"][1] for s in m._last_epoch_stats)
        upd = sum(s["counters"][2] for s in m._last_epoch_stats)
        res.append(evaluate(m) + (ms, draws, upd))
    r = np.array(res)
    print("hip update_mode=%d max_waves=%-5d p@10 train %.4f test %.4f (std %.4f, sem %.4f, n=%d)  kernel %.2f ms/epoch  %.3g inter/s  draws/inter %.2f upd/inter %.2f" % (
        um, cap, r[:, 0].mean(), r[:, 1].mean(), r[:, 1].std(), r[:, 1].std() / np.sqrt(len(r)), len(r), r[:, 2].mean() / epochs,
        train.nnz * epochs / (r[:, 2].mean() / 1e3), r[:, 3].mean() / (train.nnz * epochs),
        r[:, 4].mean() / (train.nnz * epochs)), flush=True)
