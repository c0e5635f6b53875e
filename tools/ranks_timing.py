"""Wall time of predict_rank / precision_at_k at the ML-20M shape on a subset of users."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from lightfm_amd import LightFM, synthetic
from lightfm_amd.evaluation import precision_at_k
n_eval = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
data = synthetic.named("ml-20m")
train, test = synthetic.train_test_split(data, 0.1, seed=1)
users = np.sort(np.random.RandomState(0).choice(data.shape[0], size=n_eval, replace=False))
mask = np.zeros(data.shape[0], bool); mask[users] = True
keep = mask[test.row]
test_sub = sp.coo_matrix((test.data[keep], (test.row[keep], test.col[keep])), shape=test.shape, dtype=np.float32).tocsr()
m = LightFM(no_components=64, loss="warp", random_state=1).fit(train, epochs=1)
train_csr = train.tocsr()
for _ in range(2):
    t = time.time(); p = precision_at_k(m, test_sub, train_interactions=train_csr, k=10).mean(); dt = time.time() - t
    print("precision_at_k over %d users x %d items: %.2fs (%.1f M user-item scores/s), p@10 %.4f" % (
        n_eval, data.shape[1], dt, n_eval * data.shape[1] / dt / 1e6, p), flush=True)
