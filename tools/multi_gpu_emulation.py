"""Measures the N-GPU training semantics on ONE GPU: K device-resident sessions (one per emulated
rank, each with its row shard of the interactions, its own user rows and a replica of the item
tables) run the very schedule lightfm_amd/distributed.py runs on K GPUs -- segments of the epoch
with a merge of the item tables after each -- with the merge done by lfm_sessions_merge_local
(the arithmetic of the RCCL path, csrc/session.hip: merge_group, without RCCL).  Reports
precision@10 of the merged model against one replica (K = 1) on ML-20M-shaped data.

    python tools/multi_gpu_emulation.py CONFIG [CONFIG ...]
      CONFIG = K:mode:merge_k:merge_min:merge_max      e.g. 8:adagrad:4:16384:0   1:sum:4:16384:0
    env: EMU_EPOCHS (5), EMU_SEEDS (1,2,3), EMU_EVAL_USERS (4000), EMU_SCALE (1.0)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

from lightfm_amd import LightFM, synthetic
from lightfm_amd._lightfm_fast import CSRMatrix, FastLightFM, make_opts
from lightfm_amd.distributed import MergePolicy, local_shard, merge_schedule, segment_positions
from lightfm_amd.evaluation import precision_at_k
from lightfm_amd.lightfm import _Session, _WEIGHTS

epochs = int(os.environ.get("EMU_EPOCHS", "5"))
seeds = [int(x) for x in os.environ.get("EMU_SEEDS", "1,2,3").split(",")]
n_eval = int(os.environ.get("EMU_EVAL_USERS", "4000"))
scale = float(os.environ.get("EMU_SCALE", "1.0"))
D = 64

t0 = time.time()
data = synthetic.named("ml-20m", scale=scale)
train, test = synthetic.train_test_split(data, 0.1, seed=1)
users = np.sort(np.random.RandomState(0).choice(data.shape[0], size=n_eval, replace=False))
mask = np.zeros(data.shape[0], bool)
mask[users] = True
keep = mask[test.row]
test_sub = sp.coo_matrix((test.data[keep], (test.row[keep], test.col[keep])), shape=test.shape,
                         dtype=np.float32).tocsr()
train_csr = train.tocsr()
n_users, n_items = data.shape
print("data %s train %d (%.0fs)" % (data.shape, train.nnz, time.time() - t0), flush=True)


def run(K, policy, seed):
    rng = np.random.RandomState(seed)
    model = LightFM(no_components=D, loss="warp", random_state=seed)
    model._initialize(D, n_items, n_users)
    item_f = CSRMatrix(sp.identity(n_items, dtype=np.float32, format="csr"))
    sessions, structs, shards = [], [], []
    bounds = None
    for r in range(K):
        shard, bounds = local_shard(train, r, K, bounds=bounds, rebase=True)
        b0, b1 = int(bounds[r]), int(bounds[r + 1])
        arrays = []
        for name in _WEIGHTS:
            a = getattr(model, name)
            # every replica gets its OWN copy of the item tables and a view of its user rows
            arrays.append(a[b0:b1] if name.startswith("user") else a.copy())
        st = FastLightFM(*arrays, D, 0, model.learning_rate, model.rho, model.epsilon, model.max_sampled)
        s = _Session(st, item_f, CSRMatrix(sp.identity(b1 - b0, dtype=np.float32, format="csr")))
        s.set_interactions(None, np.ascontiguousarray(shard.row), np.ascontiguousarray(shard.col),
                           shard.data, shard.data)
        s.build_positives(b1 - b0, n_items)
        s.merge_begin(1)
        sessions.append(s)
        structs.append(st)
        shards.append(shard)
    history, merges, kernel_ms = 0, 0, 0.0
    try:
        for e in range(epochs):
            for s in sessions:
                s.device_shuffle(int(rng.randint(1 << 30)), int(rng.randint(1 << 30)))
            sd = [np.array([rng.randint(1 << 30)], np.uint32) for _ in sessions]
            fr = merge_schedule(history, train.nnz, K, policy)
            pos = [segment_positions(fr, sh.nnz) for sh in shards]
            for j in range(len(fr) - 1):
                seg_ms = 0.0
                for r, s in enumerate(sessions):
                    opts, _ = make_opts()
                    opts.history = (history + int(round(train.nnz * fr[j]))) // K
                    opts.pos_begin, opts.pos_end = int(pos[r][j]), int(pos[r][j + 1])
                    if pos[r][j + 1] > pos[r][j]:
                        s.epoch("warp", 0.0, 0.0, 5, 10, sd[r], opts)
                        seg_ms = max(seg_ms, float(opts.kernel_ms))
                kernel_ms += seg_ms  # ranks run concurrently on real hardware
                if K > 1:
                    _Session.merge_local(sessions, 1, policy.mode_id())
                merges += 1
            history += train.nnz
        for r, s in enumerate(sessions):
            s.sync_to_host(structs[r])
        for name in _WEIGHTS:  # replica 0's merged item tables are THE item tables
            if not name.startswith("user"):
                getattr(model, name)[...] = getattr(structs[0], FastLightFM._names[_WEIGHTS.index(name)])
    finally:
        for s in sessions:
            s.close()
    p = precision_at_k(model, test_sub, train_interactions=train_csr, k=10).mean()
    return p, merges / float(epochs), kernel_ms / epochs


for spec in sys.argv[1:]:
    K, mode, mk, mmin, mmax = spec.split(":")
    policy = MergePolicy(merge_k=int(mk), merge_min=int(mmin), merge_max=int(mmax), mode=mode)
    res = []
    t = time.time()
    for seed in seeds:
        p, mpe, kms = run(int(K), policy, seed)
        res.append(p)
        print("  %s seed %d: p@10 test %.4f  (%.1f merges/epoch, max-over-ranks kernel %.1f ms/epoch)"
              % (spec, seed, p, mpe, kms), flush=True)
    print("K=%s mode=%s merge_k=%s min=%s max=%s: p@10 test %.4f (std %.4f)  %.1f merges/epoch  [%.0fs]"
          % (K, mode, mk, mmin, mmax, np.mean(res), np.std(res), mpe, time.time() - t), flush=True)
