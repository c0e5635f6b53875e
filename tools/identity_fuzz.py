"""Randomised sequential-parity runs of the identity BPR / logistic kernels (lane-group kernels d <= 12, the tile kernel's BPR / logistic instantiations above, with and
without an L2 penalty): random shapes, densities, widths, labels with zeros, sample weights; one interaction per launch against the oracle -- negatives, draw counts,
counters exact, arrays within the bar of float-atomic publication.  Mode "frozen" (BPR): sample_weight = 0 at FULL concurrency -- a few launches per epoch, the whole grid
or two workgroups walking many passes, problems of up to 60 000 interactions: every negative, draw count and counter exact, no array moves.
    python tools/identity_fuzz.py [cases=150] [seed=1] [sequential|frozen]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from oracle import oracle
from oracle.oracle import ARRAYS
from tests import helpers as H
import lightfm_amd._lightfm_fast as fast
from lightfm_amd.options import options

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
Cm = fast.CSRMatrix
bad, flags_seen = 0, {}
frozen = len(sys.argv) > 3 and sys.argv[3] == "frozen"
for case in range(cases if frozen else 0):
    d = int(rng.choice([4, 8, 10, 12, 16, 30, 64, 100, 200]))
    nu, ni = int(rng.randint(50, 4000)), int(rng.randint(20, 800))
    coo = H.make_interactions(nu, ni, int(rng.randint(500, 60000)), seed=int(rng.randint(1 << 30)), ratings=True, zipf=float(rng.choice([0.3, 0.7, 1.0])))
    st = oracle.State(ni, nu, d, rng, max_sampled=10)
    a, b = st.copy(), st.copy()
    zeros = np.zeros_like(coo.data)
    shuffle, seeds = H.epoch_inputs(coo, rng)
    options.set(mode="parallel", log_samples=True, launches_per_epoch=int(rng.randint(1, 6)), ramp_k=-1, max_waves=int(rng.choice([0, 8, 64])), update_mode=0,
                warp_kernel=0, feat_kernel=0, first_batch=0, debug=0)
    item_f, user_f, pos = H.identity_features(ni), H.identity_features(nu), H.positives_csr(coo)
    fl = fast.FastLightFM(*a.arrays(), a.d, 0, a.lr, a.rho, a.eps, a.max_sampled)
    fast.fit_bpr(Cm(item_f), Cm(user_f), Cm(pos), coo.row, coo.col, coo.data, zeros, shuffle, fl, 0.05, 0.0, 0.0, len(seeds), H.FixedRandom(seeds))
    o = oracle.Opts(len(shuffle), rng_mode=1, log=True)
    oracle.fit_bpr(item_f, user_f, pos, coo.row, coo.col, coo.data, zeros, shuffle, b, 0.0, 0.0, seeds, o)
    neg, sampled = options.last_logs
    key = ("bpr frozen", int(options.last_kernel_used), int(options.last_plan_flags) & (256 | 512 | 1024 | 2048))
    flags_seen[key] = flags_seen.get(key, 0) + 1
    ok = np.array_equal(neg, o.neg) and np.array_equal(sampled, o.sampled) and list(options.last_counters) == list(o.counters)
    ok = ok and all(np.array_equal(getattr(a, n), getattr(st, n)) for n in ARRAYS)
    if not ok:
        bad += 1
        print("frozen case %d FAILED: d=%d %dx%d nnz=%d kernel %s" % (case, d, nu, ni, coo.nnz, key), flush=True)
for case in range(0 if frozen else cases):
    loss = "bpr" if rng.rand() < 0.5 else "logistic"
    d = int(rng.choice([1, 3, 4, 8, 10, 12, 13, 16, 24, 31, 32, 50, 64, 65, 100, 128, 130, 200, 256]))
    nu, ni = int(rng.randint(1, 40)), int(rng.randint(2, 60))
    dens = float(rng.choice([0.05, 0.2, 0.5, 0.9, 1.0]))
    dense = rng.rand(nu, ni) < dens
    dense[rng.randint(nu), rng.randint(ni)] = True
    m = sp.coo_matrix(dense.astype(np.float32))
    keep = rng.permutation(m.nnz)[: int(rng.randint(1, min(m.nnz, 300) + 1))]
    rows, cols = m.row[keep].astype(np.int32), m.col[keep].astype(np.int32)
    if loss == "bpr":
        vals = (1.0 + rng.rand(len(keep))).astype(np.float32)
        vals[rng.rand(len(keep)) < 0.1] = 0.0
        w = vals if rng.rand() < 0.5 else (0.25 + rng.rand(len(keep)) * 1.5).astype(np.float32)
    else:
        vals = np.where(rng.rand(len(keep)) < 0.5, 1.0, -1.0).astype(np.float32)
        vals[rng.rand(len(keep)) < 0.05] = 0.0
        w = (0.25 + rng.rand(len(keep)) * 1.5).astype(np.float32)
    coo = sp.coo_matrix((vals, (rows, cols)), shape=(nu, ni), dtype=np.float32)
    alpha = float(rng.choice([0.0, 0.0, 1e-3]))
    st = oracle.State(ni, nu, d, rng, max_sampled=10)
    st.item_embeddings *= 4 * np.sqrt(d); st.user_embeddings *= 4 * np.sqrt(d)
    st.item_biases[:] = rng.randn(ni).astype(np.float32) * 0.3
    st.user_biases[:] = rng.randn(nu).astype(np.float32) * 0.3
    a, b = st.copy(), st.copy()
    item_f, user_f = H.identity_features(ni), H.identity_features(nu)
    options.set(mode="parallel", log_samples=(loss == "bpr"), launches_per_epoch=coo.nnz, update_mode=0, warp_kernel=0, feat_kernel=0, first_batch=0, max_waves=0, debug=0, ramp_k=0)
    ok, why = True, ""
    for ep in range(2):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        fl = fast.FastLightFM(*a.arrays(), a.d, 0, a.lr, a.rho, a.eps, a.max_sampled)
        o = oracle.Opts(len(shuffle), rng_mode=1, log=True)
        if loss == "bpr":
            pos = H.positives_csr(coo)
            fast.fit_bpr(Cm(item_f), Cm(user_f), Cm(pos), coo.row, coo.col, coo.data, w, shuffle, fl, 0.05, alpha, 2 * alpha, len(seeds), H.FixedRandom(seeds))
            oracle.fit_bpr(item_f, user_f, pos, coo.row, coo.col, coo.data, w, shuffle, b, alpha, 2 * alpha, seeds, o)
            neg, sampled = options.last_logs
            if not (np.array_equal(neg, o.neg) and np.array_equal(sampled, o.sampled)):
                ok, why = False, "samples"
        else:
            fast.fit_logistic(Cm(item_f), Cm(user_f), coo.row, coo.col, coo.data, w, shuffle, fl, 0.05, alpha, 2 * alpha, 1)
            oracle.fit_logistic(item_f, user_f, coo.row, coo.col, coo.data, w, shuffle, b, alpha, 2 * alpha, o)
        if list(options.last_counters) != list(o.counters):
            ok, why = False, why + " counters"
    key = (loss, int(options.last_kernel_used), int(options.last_plan_flags) & (256 | 512 | 1024 | 2048))
    flags_seen[key] = flags_seen.get(key, 0) + 1
    worst = 0.0
    for n in ARRAYS:
        x, y = getattr(a, n).astype(np.float64), getattr(b, n).astype(np.float64)
        if x.size:
            worst = max(worst, float(np.max(np.abs(x - y) / (5e-6 + 5e-5 * np.abs(y)))))
    if worst > 1.0:
        ok, why = False, why + " arrays x%.1f" % worst
    if not ok:
        bad += 1
        print("case %d FAILED (%s): %s d=%d %dx%d nnz=%d alpha=%g kernel %s" % (case, why.strip(), loss, d, nu, ni, coo.nnz, alpha, key), flush=True)
print("%d cases, %d failed; (loss, kernel_used, plan-flag bits 8-11) -> cases: %s" % (cases, bad, sorted(flags_seen.items())))
