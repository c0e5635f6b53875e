"""(with "logistic" as third argument: LightFM()'s default model -- logistic, no_components = 10 -- and d = 64, on the train positives plus as many uniformly drawn
explicit negatives.)  precision@10 at the FULL ML-20M shape for identity BPR models (the tile kernel's BPR instantiation at d = 64, the lane-group kernel at d = 10): this backend in
its shipped mode against the reference's OpenMP build (16 threads), all test users, the reference's metric.    python tools/quality20m_identity.py [epochs=3] [seeds=1,2] [bpr|logistic] [widths]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from lightfm_amd import LightFM, synthetic
from lightfm_amd.evaluation import precision_at_k
from oracle.ref_model import RefLightFM

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
seeds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2]
loss = sys.argv[3] if len(sys.argv) > 3 else "bpr"
data = synthetic.named("ml-20m")
train, test = synthetic.train_test_split(data, 0.05, seed=1)
tr, te = train.tocsr(), test.tocsr()
fit_on = train
if loss == "logistic":
    from tests.test_precision_parity import _with_explicit_negatives
    fit_on = _with_explicit_negatives(train, test)
p10 = lambda m: float(precision_at_k(m, te, train_interactions=tr, k=10).mean())
print("# %s; ML-20M shape %s, %d train / %d test interactions, %d epochs, seeds %s; precision@10 over all %d test users" % (loss + (" on %d labelled interactions" % fit_on.nnz if loss == "logistic" else ""), data.shape, train.nnz, test.nnz, epochs, seeds, len(np.unique(test.row))), flush=True)
widths = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else ([64, 10] if loss == "bpr" else [10, 64])
for d in widths:
    def fit_ref(seed):
        t = time.time()
        r = RefLightFM(no_components=d, loss=loss, random_state=seed)
        r.fit(fit_on, epochs=epochs, num_threads=min(16, os.cpu_count() or 1))
        return r, time.time() - t
    with ThreadPoolExecutor(max_workers=2) as pool:
        pending = [pool.submit(fit_ref, s) for s in seeds]
        hip = []
        for s in seeds:
            t = time.time()
            m = LightFM(no_components=d, loss=loss, random_state=s)
            m.fit(fit_on, epochs=epochs)
            dt = time.time() - t
            st = m._last_epoch_stats[-1]
            hip.append(p10(m))
        print(loss + " d=%-3d hip %s  mean %.5f  (fit %.2f s, kernel_used %d, plan flags %d, in flight %d)" % (d, [round(x, 5) for x in hip], np.mean(hip), dt, st["kernel_used"], st["plan_flags"], st["in_flight"]), flush=True)
        ref, secs = [], []
        for f in pending:
            r, sec = f.result()
            ref.append(p10(r)); secs.append(sec)
        print(loss + " d=%-3d ref %s  mean %.5f  (fit %.0f s at 16 threads)   delta %+.5f" % (d, [round(x, 5) for x in ref], np.mean(ref), np.mean(secs), np.mean(hip) - np.mean(ref)), flush=True)
