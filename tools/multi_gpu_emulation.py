"""Measures the N-GPU training semantics on ONE GPU: K device-resident sessions (one per emulated
rank, each with its row shard of the interactions, its own user rows and a replica of the item
tables) run the very schedule lightfm_amd/distributed.py runs on K GPUs -- segments of the epoch
with a merge of the item tables after each -- with the merge done by lfm_sessions_merge_local /
lfm_sessions_merge_local_sparse (the arithmetic of the RCCL path, csrc/session.hip: merge_group /
merge_group_sparse, without RCCL; "overlap" applies every exchange one merge late, as the overlapped
multi-GPU exchange does).  Reports precision@10 of the merged model against one replica (K = 1).

    python tools/multi_gpu_emulation.py CONFIG [CONFIG ...]
      CONFIG = K:mode:merge_k:merge_min:merge_max[:flavour[:rows_k]]
                flavour = dense | sparse | overlap (default) | late<E> (sparse, overlapped only once the model has
                seen E epochs: late1, late0.5); a "+hot" suffix merges the hot rows (MergePolicy.hot_share) every
                K * 2**17 interactions between the full merges      rows_k = MergePolicy.rows_k (default 0 = off)
                e.g. 8:adagrad:4:16384:0   1:sum:4:16384:0:dense   8:adagrad:4:16384:0:late1:16
    env: EMU_SHAPE  c2 (ML-20M shape, WARP d=64, identity; default) | c3 (ML-20M shape, BPR d=128, item
                    features [identity | 8 tags of 1128]: shared rows in the replicated tables) |
                    c4s (a 1/8-scale C4: 156 k users x 625 k items x 62.5 M interactions with latent
                    structure, WARP d=64: the table-scaled merge intervals of MergePolicy.rows_k)
         EMU_EPOCHS (5), EMU_SEEDS (1,2,3), EMU_EVAL_USERS (4000), EMU_SCALE (1.0)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

from lightfm_amd import LightFM, synthetic
from lightfm_amd._lightfm_fast import CSRMatrix, FastLightFM, make_opts
from lightfm_amd.distributed import MergePolicy, hot_rows, local_shard, merge_plan, segment_positions
from lightfm_amd.evaluation import precision_at_k
from lightfm_amd.lightfm import _Session, _WEIGHTS

epochs = int(os.environ.get("EMU_EPOCHS", "5"))
seeds = [int(x) for x in os.environ.get("EMU_SEEDS", "1,2,3").split(",")]
n_eval = int(os.environ.get("EMU_EVAL_USERS", "4000"))
scale = float(os.environ.get("EMU_SCALE", "1.0"))
shape = os.environ.get("EMU_SHAPE", "c2")
D, LOSS = (128, "bpr") if shape == "c3" else (64, "warp")

t0 = time.time()
if shape == "c4s":
    data = synthetic.make_interactions(156_250, 625_000, int(62_500_000 * scale), seed=42, n_clusters=256)
else:
    data = synthetic.named("ml-20m", scale=scale)
feats = synthetic.tag_item_features(data.shape[1]) if shape == "c3" else None
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


n_item_feat = feats.shape[1] if feats is not None else n_items


def run(K, policy, seed, flavour):
    use_hot = flavour.endswith("+hot")
    flavour = flavour[:-4] if use_hot else flavour
    hot = hot_rows(feats, policy.hot_share) if use_hot else []
    rng = np.random.RandomState(seed)
    model = LightFM(no_components=D, loss=LOSS, random_state=seed)
    model._initialize(D, n_item_feat, n_users)
    item_f = CSRMatrix(feats if feats is not None else sp.identity(n_items, dtype=np.float32, format="csr"))
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
        if len(hot):
            s.set_hot_rows(0, hot)
        sessions.append(s)
        structs.append(st)
        shards.append(shard)
    history, merges, kernel_ms = 0, 0, 0.0
    try:
        for e in range(epochs):
            for s in sessions:
                s.device_shuffle(int(rng.randint(1 << 30)), int(rng.randint(1 << 30)))
            sd = [np.array([rng.randint(1 << 30)], np.uint32) for _ in sessions]
            fr, kinds = merge_plan(history, train.nnz, K, policy, n_item_feat, len(hot) > 0)
            pos = [segment_positions(fr, sh.nnz) for sh in shards]
            for j in range(len(fr) - 1):
                seg_ms = 0.0
                for r, s in enumerate(sessions):
                    opts, _ = make_opts()
                    opts.history = (history + int(round(train.nnz * fr[j]))) // K
                    opts.pos_begin, opts.pos_end = int(pos[r][j]), int(pos[r][j + 1])
                    if pos[r][j + 1] > pos[r][j]:
                        s.epoch(LOSS, 0.0, 0.0, 5, 10, sd[r], opts)
                        seg_ms = max(seg_ms, float(opts.kernel_ms))
                kernel_ms += seg_ms  # ranks run concurrently on real hardware
                if K > 1:
                    seen = history + train.nnz * fr[j + 1]
                    ov = flavour == "overlap" or (flavour.startswith("late") and seen >= float(flavour[4:]) * train.nnz)
                    if flavour == "dense":
                        _Session.merge_local(sessions, 1, policy.mode_id())
                    elif kinds[j] == "hot":
                        _Session.merge_local_hot(sessions, 1, policy.mode_id(), overlap=ov)
                    else:
                        _Session.merge_local_sparse(sessions, 1, policy.mode_id(), overlap=ov)
                merges += 1
            if K > 1 and flavour != "dense":
                _Session.merge_local_flush(sessions)  # as DistributedFit.run_epoch does at the end of an epoch
            history += train.nnz
        for r, s in enumerate(sessions):
            s.sync_to_host(structs[r])
        for name in _WEIGHTS:  # replica 0's merged item tables are THE item tables
            if not name.startswith("user"):
                getattr(model, name)[...] = getattr(structs[0], FastLightFM._names[_WEIGHTS.index(name)])
    finally:
        for s in sessions:
            s.close()
    p = precision_at_k(model, test_sub, train_interactions=train_csr, k=10, item_features=feats).mean()
    return p, merges / float(epochs), kernel_ms / epochs


for spec in sys.argv[1:]:
    parts = spec.split(":")
    K, mode, mk, mmin, mmax = parts[:5]
    flavour = parts[5] if len(parts) > 5 else "overlap"
    policy = MergePolicy(merge_k=int(mk), merge_min=int(mmin), merge_max=int(mmax), mode=mode)
    if len(parts) > 6:
        policy.rows_k = int(parts[6])
    res = []
    t = time.time()
    for seed in seeds:
        p, mpe, kms = run(int(K), policy, seed, flavour)
        res.append(p)
        print("  %s seed %d: p@10 test %.4f  (%.1f merges/epoch, max-over-ranks kernel %.1f ms/epoch)"
              % (spec, seed, p, mpe, kms), flush=True)
    print("%s K=%s mode=%s merge_k=%s min=%s max=%s %s rows_k=%d: p@10 test %.4f (std %.4f)  %.1f merges/epoch  [%.0fs]"
          % (shape, K, mode, mk, mmin, mmax, flavour, policy.rows_k, np.mean(res), np.std(res), mpe, time.time() - t), flush=True)
