"""precision@10 with and without the plain-store user rows (lfm_opts.user_store; options.debug 4096 switches it off) at the FULL
C2 / C3 shapes, N seeds per arm, all users with test interactions (the precision gates of tests/ run scaled problems whose user
counts switch the feature off by the session's own rule).

    python tools/ustore_quality.py [c2|c3] [seeds=6] [epochs=3]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from lightfm_amd import LightFM, options, synthetic
from lightfm_amd.evaluation import precision_at_k

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
data = synthetic.make_interactions(138493, 26744, 21000000, seed=42)
train, test = synthetic.split_off_test(data, 20000263, seed=1)
feats = synthetic.tag_item_features(26744) if cfg == "c3" else None
loss, d = ("bpr", 128) if cfg == "c3" else ("warp", 64)
tr, te = train.tocsr(), test.tocsr()
res = {}
for arm, debug in (("user rows stored", 0), ("user rows by atomics", 4096)):
    p = []
    for seed in range(1, n_seeds + 1):
        options.set(mode="parallel", debug=debug)
        m = LightFM(no_components=d, loss=loss, random_state=seed)
        m.fit(train, item_features=feats, epochs=epochs)
        p.append(float(precision_at_k(m, te, train_interactions=tr, k=10, item_features=feats).mean()))
    options.set(debug=0)
    res[arm] = np.array(p)
    print("%s %-22s p@10 %.5f (s.e. %.5f, n=%d)  %s" % (cfg, arm, res[arm].mean(), res[arm].std(ddof=1) / np.sqrt(n_seeds), n_seeds,
                                                         " ".join("%.5f" % x for x in p)), flush=True)
print("%s: stored - atomics = %+.5f" % (cfg, res["user rows stored"].mean() - res["user rows by atomics"].mean()))
