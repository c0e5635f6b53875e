"""Emulates the N-GPU training semantics on ONE GPU: K replicas train a slice of their user shard
from the same tables (sequentially, each on the device), then the tables are merged as
csrc/session.hip does with RCCL: X := X_start + sum_k (X_k - X_start) -- S times per epoch.
Reports precision@10 of the merged model (ML-20M-shaped data; reference 0.1766, one replica
0.1774).

    python tools/multi_gpu_emulation.py [K=8] [epochs=5] [syncs_per_epoch=1] [merge=sum|mean] [n_eval=4000]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from lightfm_amd import LightFM, synthetic
from lightfm_amd._lightfm_fast import CSRMatrix, make_opts
from lightfm_amd.distributed import local_shard
from lightfm_amd.evaluation import precision_at_k
from lightfm_amd.lightfm import _Session, _WEIGHTS

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1
merge = sys.argv[4] if len(sys.argv) > 4 else "sum"
n_eval = int(sys.argv[5]) if len(sys.argv) > 5 else 4000
seeds = [int(x) for x in os.environ.get("EMU_SEEDS", "1,2,3").split(",")]
data = synthetic.named("ml-20m")
train, test = synthetic.train_test_split(data, 0.1, seed=1)
users = np.sort(np.random.RandomState(0).choice(data.shape[0], size=n_eval, replace=False))
mask = np.zeros(data.shape[0], bool); mask[users] = True
keep = mask[test.row]
test_sub = sp.coo_matrix((test.data[keep], (test.row[keep], test.col[keep])), shape=test.shape, dtype=np.float32).tocsr()
train_csr = train.tocsr()
n_users, n_items = data.shape
item_f = CSRMatrix(sp.identity(n_items, dtype=np.float32, format="csr"))
user_f = CSRMatrix(sp.identity(n_users, dtype=np.float32, format="csr"))
shards = [local_shard(train, r, K)[0] for r in range(K)]
positives = [CSRMatrix(s.tocsr().sorted_indices()) for s in shards]
print("K=%d replicas, %d syncs per epoch, merge=%s" % (K, S, merge), flush=True)

res = []
for seed in seeds:
    rng = np.random.RandomState(seed)
    model = LightFM(no_components=64, loss="warp", random_state=seed)
    model._initialize(64, n_items, n_users)
    state = {n: getattr(model, n).copy() for n in _WEIGHTS}
    history = 0
    for e in range(epochs):
        order = [rng.permutation(s.nnz) for s in shards]
        for seg in range(S):
            total = {n: np.zeros_like(v) for n, v in state.items()}
            seg_n = 0
            for r in range(K):
                idx = np.sort(order[r][seg * shards[r].nnz // S:(seg + 1) * shards[r].nnz // S])
                rows = np.ascontiguousarray(shards[r].row[idx]); cols = np.ascontiguousarray(shards[r].col[idx])
                vals = np.ascontiguousarray(shards[r].data[idx])
                for n in _WEIGHTS:
                    setattr(model, n, state[n].copy())
                struct = model._get_lightfm_data()
                sess = _Session(struct, item_f, user_f)
                try:
                    sess.set_interactions(positives[r], rows, cols, vals, vals)
                    sess.device_shuffle(int(rng.randint(1 << 30)), int(rng.randint(1 << 30)))
                    opts, _ = make_opts()
                    opts.history = history // K
                    sess.epoch("warp", 0.0, 0.0, 5, 10, np.array([rng.randint(1 << 30)], np.uint32), opts)
                    sess.sync_to_host(struct)
                finally:
                    sess.close()
                for n in _WEIGHTS:
                    total[n] += getattr(model, n) - state[n]
                seg_n += len(idx)
            history += seg_n
            for n in _WEIGHTS:
                scale = (1.0 / K) if (merge == "mean" and n.startswith("item") and "gradients" not in n) else 1.0
                state[n] = (state[n] + scale * total[n]).astype(np.float32)
    for n in _WEIGHTS:
        setattr(model, n, state[n])
    p = precision_at_k(model, test_sub, train_interactions=train_csr, k=10).mean()
    res.append(p)
    print("  seed %d: p@10 test %.4f" % (seed, p), flush=True)
print("K=%d syncs/epoch=%d merge=%s: p@10 test %.4f (std %.4f)" % (K, S, merge, np.mean(res), np.std(res)), flush=True)
