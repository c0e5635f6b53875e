#!/usr/bin/env python
"""bench.py -- positive interactions/sec/epoch of the epoch kernels on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4shard|c5shard]
                    [--scaling strong|weak]

Workloads (BASELINE.json `configs`, synthetic data of the named shapes -- no dataset is reachable
offline; lightfm_amd/synthetic.py):

  c2 (default)  MovieLens-20M shape (138,493 x 26,744 x 20,000,263), loss=warp, no_components=64,
                identity features                                     [BASELINE configs[1], the metric]
  c3            the same interactions, loss=bpr, no_components=128, item features = [identity | 8 tags
                of 1,128]                                              [configs[2]]
  c4shard       one GPU's row shard of configs[3]: 1.25 M users x 5 M items x 62.5 M interactions,
                warp, no_components=64, the FULL 5 M-row item tables (2.6 GB: genuinely HBM-bound)
  c5shard       one GPU's row shard of configs[4]: 6.25 M users x 10 M items x 250 M interactions,
                warp-kos (k=5, n=10), no_components=128, item-feature CSR of 1 M embedding rows, avg 8 nnz

A "step" is `epochs_per_step` EPOCHS of ONE continuing training run (reported in `config`; chosen
during warm-up so that the K timed steps last >= ~6 s): every epoch is what LightFM.fit_partial
does per epoch -- the keyed on-device shuffle, the kernel seeds, one pass of the hot path over all
interactions through the C ABI (include/lfm_hip.h: lfm_session_epoch), the on-device finite
check.  Inputs (weights, COO, positives lookup) are resident in HBM before the timed region.

N > 1 (launched by torch.distributed.run, one rank per GPU).  c2 / c3 default to STRONG scaling:
ONE ML-20M-shaped COO sharded row-wise (by user) over the N ranks, user tables partitioned (a
rank allocates only its own users' rows), item tables replicated and merged by RCCL all-reduce
of their deltas at the cadence of lightfm_amd/distributed.py (merge_schedule), inside the timed
region; `--scaling weak` gives every rank its own full-size shard instead.  c4shard / c5shard
are per-GPU shards by definition (weak).  torch.distributed (gloo) is used only to hand the RCCL
unique id to the ranks, for the barriers and for the max-over-ranks of the elapsed time.

Besides the contract fields the JSON line carries `roofline` (dominant kernel, algorithmic bytes
per SURVEY.md 8(d), HIP-event launch times), `cpu_baseline` (the reference's own compiled
Cython/OpenMP path on this box's host cores, N = 1 only), `quality` (precision@10 of this
backend and of the reference trained on the same data, N = 1, c2 / c3) and `end_to_end_fit`
(LightFM.fit through the public API, uploads and downloads included).

At N = 1 with no --config the line also carries `extra_configs`: short timed legs of the other
BASELINE shapes (c3, c4shard, and c5shard at its full per-GPU size) run after c2 in the same
process, each with its own `value` and `roofline` (--no-extra skips them; --extra picks).
`roofline.traffic` is the HBM traffic per launch of the dominant kernel taken from the committed
rocprofv3 --pmc summary of the same command (profiles/traffic.json names the source file per
config and kernel; counters cannot be collected inside this run), null when the committed
summary is of another kernel.

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md:35
ATOMIC_PEAK_GOPS = 320.0  # global_atomic_add_f32 lanes per second, measured (tools/membench.hip)
MAX_SAMPLED = 10
KNOBS = ("update_mode", "first_batch", "launches_per_epoch", "max_waves", "warp_kernel", "feat_kernel", "debug",
         "ramp_k", "shared_cap")

CONFIGS = {
    "c2": dict(loss="warp", d=64, shape="ml-20m", features=None, default_scaling="strong",
               label="MovieLens-20M shape (138493 users x 26744 items x %d interactions), loss=warp, "
                     "no_components=64, identity features, adagrad, max_sampled=10"),
    "c3": dict(loss="bpr", d=128, shape="ml-20m", features="tags", default_scaling="strong",
               label="MovieLens-20M shape (138493 users x 26744 items x %d interactions), loss=bpr, "
                     "no_components=128, item features [identity | 8 tags of 1128] (9 nnz/row), adagrad"),
    "c4shard": dict(loss="warp", d=64, shape=(1_250_000, 5_000_000, 62_500_000), features=None,
                    default_scaling="weak",
                    label="one GPU's row shard of 10M users x 5M items x 500M interactions: 1.25M users x 5M "
                          "items x %d interactions, loss=warp, no_components=64, identity features, full "
                          "5M-row item tables"),
    "c5shard": dict(loss="warp-kos", d=128, shape=(6_250_000, 10_000_000, 250_000_000), features="hashed",
                    default_scaling="weak",
                    label="one GPU's row shard of 50M users x 10M items x 2B interactions: 6.25M users x 10M "
                          "items x %d interactions, loss=warp-kos (k=5, n=10), no_components=128, item-feature "
                          "CSR over 1M embedding rows, avg 8 nnz/row"),
}


def rep_bytes(f, d):
    """R(f) of SURVEY.md 8(d): indptr pair + (index, value) entries + embedding and bias rows."""
    return 8 + 8 * f + 4 * f * (d + 1)


def update_bytes(f, d):
    """Up(f): read W, G + write W, G per cell of the f feature rows of one representation."""
    return 16 * f * (d + 1)


def algorithmic_bytes(loss, c, d, f_u, f_i, mean_probe_bytes, n_examples, kos_n=10, mean_kos_pos=None):
    """SURVEY.md 8(d), per loss.  c = (positives visited, draws, updates, in_positives probes)."""
    npos, draws, updates, probes = [float(x) for x in c]
    R, Up = rep_bytes, update_bytes
    if loss == "warp":
        return (npos * (20 + R(f_u, d) + R(f_i, d)) + draws * R(f_i, d)
                + updates * (Up(f_u, d) + 2 * Up(f_i, d)) + probes * mean_probe_bytes)
    if loss == "bpr":
        return npos * (20 + R(f_u, d) + 2 * R(f_i, d) + Up(f_u, d) + 2 * Up(f_i, d)) + probes * mean_probe_bytes
    if loss == "logistic":
        return n_examples * (20 + R(f_u, d) + R(f_i, d) + Up(f_u, d) + Up(f_i, d))
    # warp-kos: 8 B of COO, min(n, len_u) sampled positives + the chosen one re-read + S negatives
    kpos = mean_kos_pos if mean_kos_pos is not None else kos_n
    return (npos * (8 + R(f_u, d) + (kpos + 1) * R(f_i, d)) + draws * R(f_i, d)
            + updates * (Up(f_u, d) + 2 * Up(f_i, d)) + probes * mean_probe_bytes)


def build_workload(name, rank, world, scaling, scale, want_test):
    """(this rank's training COO with LOCAL user ids, item feature CSR or None, test COO or None,
    global interaction count, users of the rank, n_items)."""
    from lightfm_amd import synthetic
    from lightfm_amd.distributed import local_shard
    cfg = CONFIGS[name]
    test = None
    if cfg["shape"] == "ml-20m":
        n_users, n_items, nnz = synthetic.SHAPES["ml-20m"]
        nnz = int(nnz * scale)
        seed = 42 if scaling == "strong" else 42 + rank
        n_test = int(nnz * 0.05) if want_test else 0
        data = synthetic.make_interactions(n_users, n_items, nnz + n_test, seed=seed)
        if n_test:
            data, test = synthetic.split_off_test(data, min(nnz, data.nnz - 1), seed=1)
        if scaling == "strong" and world > 1:
            train, _ = local_shard(data, rank, world, rebase=True)
            global_n = data.nnz
        else:
            train, global_n = data, data.nnz * world
    else:
        n_users, n_items, nnz = cfg["shape"]
        train = synthetic.big_interactions(n_users, n_items, int(nnz * scale), seed=4 + rank)
        global_n = train.nnz * world
    return train, build_features(name, n_items), test, global_n, train.shape[0], n_items


def build_features(name, n_items):
    """The item feature CSR of a config (None = identity)."""
    from lightfm_amd import synthetic
    cfg = CONFIGS[name]
    if cfg["features"] == "tags":
        return synthetic.tag_item_features(n_items)
    if cfg["features"] == "hashed":
        return synthetic.hashed_item_features(n_items)
    return None


def precision_at_10(model, train, test, item_features):
    """The reference's precision_at_k (lightfm/evaluation.py:14-87), mean over ALL users with test
    interactions (the device ranks kernel scores every user x item pair in tens of milliseconds)."""
    from lightfm_amd.evaluation import precision_at_k
    return float(precision_at_k(model, test.tocsr(), train_interactions=train.tocsr(), k=10,
                                item_features=item_features).mean())


def head_users(coo, nu, n_items):
    """The first `nu` users' interactions of a COO (a row sub-sample over the full item side)."""
    keep = coo.row < nu
    return sp.coo_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=(nu, n_items), dtype=np.float32)


def thread_scaling(cfg_name, sub, feats, log, counts):
    """The reference's native epoch call at several OpenMP thread counts on one row sub-sample (SURVEY.md 8(d):
    T = 1 and T = all host cores; 16 is what `cpu_baseline.value` uses): epoch 2 of a fresh fit each."""
    out = {}
    for T in counts:
        cpu, _ = reference_leg(cfg_name, sub, None, feats, 2, log, "", threads=T)
        if cpu:
            out[str(T)] = cpu["value"]
    return out


def extra_cpu_leg(name, env, pieces):
    """cpu_baseline of an extra_configs leg: the reference (16 threads) on a bounded row sub-sample of the leg's
    own workload over the full item-side tables."""
    train, feats, n_users, n_items = pieces["train"], pieces["feats"], pieces["n_users"], pieces["n_items"]
    frac = {"c3": 16, "c4shard": 50, "c5shard": 200}.get(name, 50)
    nu = max(1000, n_users // frac)
    sub = head_users(train, nu, n_items)
    cpu, _ = reference_leg(name, sub, None, feats, 2, env.log,
                           "the first %d users' %d interactions of this workload (1/%d row sub-sample) over the full "
                           "item-side tables" % (nu, sub.nnz, frac))
    return cpu


def reference_leg(cfg_name, train, test, feats, epochs, log, sample_note, threads=None):
    """The reference's compiled Cython/OpenMP path (oracle/_ref/fast) on this box's host cores:
    throughput of its native epoch call (and with its per-epoch host prologue), and -- when a
    test set is given -- precision@10 of the model it trained."""
    from oracle import oracle
    from oracle.ref_model import RefLightFM
    if not oracle.ref_available("fast"):
        return None, None
    cfg = CONFIGS[cfg_name]
    ncpu = os.cpu_count() or 1
    # 16: where the reference's Hogwild stops scaling on these boxes (cpu_baseline.thread_scaling of every run
    # carries T = 1 / 16 / all cores measured side by side)
    threads = min(16, ncpu) if threads is None else max(1, min(int(threads), ncpu))
    m = RefLightFM(no_components=cfg["d"], loss=cfg["loss"], random_state=7, max_sampled=MAX_SAMPLED)
    m.native_seconds = []
    t0 = time.time()
    walls = []
    for _ in range(epochs):
        t1 = time.time()
        m.fit_partial(train, item_features=feats, epochs=1, num_threads=threads)
        walls.append(time.time() - t1)
    use = slice(1, None) if epochs > 1 else slice(0, None)  # epoch 0 pays the page faults
    native = float(np.mean(m.native_seconds[use]))
    wall = float(np.mean(walls[use]))
    log("cpu_baseline (%d threads): native call %.3g interactions/s, with host prologue %.3g (%.0fs)"
        % (threads, train.nnz / native, train.nnz / wall, time.time() - t0))
    cpu = {"value": train.nnz / native, "unit": "interactions/s", "cores": threads, "kind": "reference",
           "with_host_prologue": train.nnz / wall, "host_cpus": ncpu,
           "sample": sample_note + "; reference v1.17 Cython/OpenMP build (-O2 -ffast-math -march=x86-64-v3 "
                     "-fopenmp), native epoch call only (with_host_prologue adds its per-epoch tocsr + "
                     "shuffle), mean of epochs 2..%d" % epochs}
    p_ref = precision_at_10(m, train, test, feats) if test is not None else None
    return cpu, p_ref


def committed_traffic(config_name, kernel_name, interactions_per_launch=None):
    """(HBM bytes per launch, source, the profiled run's own figures) of the dominant kernel from profiles/traffic.json.
    Where the committed entry carries bytes per INTERACTION they are scaled to this run's launch length (the profiled run
    and this one differ in launch count and, slightly, in the epochs they cover; the entry's own traffic / algorithmic
    ratio is the consistent pair and is reported next to it)."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        e = table.get(config_name)
        if e and kernel_name in e.get("kernel", ""):
            if interactions_per_launch and e.get("hbm_bytes_per_interaction"):
                extra = {k: e[k] for k in ("traffic_over_algorithmic_profiled", "updates_per_interaction_profiled",
                                           "draws_per_interaction_profiled", "profiled_epochs", "profiled_scale") if k in e}
                return float(e["hbm_bytes_per_interaction"]) * interactions_per_launch, e.get("source"), extra
            return float(e["hbm_bytes_per_launch"]), e.get("source"), {}
        if e:
            return None, "profiles/traffic.json holds %s for this config, this run's kernel is %s" % (e.get("kernel"), kernel_name), {}
    except Exception as e:  # reporting only
        return None, "profiles/traffic.json unreadable: %r" % (e,), {}
    return None, "no committed counter summary for this config", {}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="default: c2, followed at N = 1 by short legs of c3 / c4shard / c5shard (extra_configs)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs legs of the default run")
    ap.add_argument("--extra", default="c3,c4shard,c5shard", help="which extra legs (comma separated)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default=None)
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the interactions (debug)")
    ap.add_argument("--emulate-shard", type=int, default=0,
                    help="debug, N = 1: run rank 0's row shard of a K-way strong-scaling split on this one GPU")
    ap.add_argument("--epochs-per-step", type=int, default=0, help="0 = calibrate so the timed region is >= ~6 s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-quality", action="store_true")
    ap.add_argument("--no-fit", action="store_true")
    ap.add_argument("--item-alpha", type=float, default=0.0, help="L2 penalty on item features (BASELINE: 0)")
    ap.add_argument("--user-alpha", type=float, default=0.0, help="L2 penalty on user features (BASELINE: 0)")
    ap.add_argument("--merge-mode", default=None, help="sum | mean | adagrad (N > 1)")
    ap.add_argument("--merge-k", type=int, default=None)
    ap.add_argument("--merge-max", type=int, default=None)
    ap.add_argument("--merge-dense", action="store_true", help="N > 1: the dense synchronous merge of round 2")
    ap.add_argument("--merge-overlap", action="store_true",
                    help="N > 1: sparse merges overlap the next segment (land one merge late; costs precision@10)")
    for knob in KNOBS:
        ap.add_argument("--" + knob.replace("_", "-"), type=int, default=None, help="backend option (tuning)")
    return ap.parse_args()


class Env(object):
    """Process-wide context of a bench run (ranks, the distributed plane, logging)."""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus and self.world > 1:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world))
        self.dist = None
        self.cache = {}  # generated workloads shared between legs (c2 and c3 train on the same COO)

    def log(self, msg):
        if self.rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)


def run_config(name, env, steps, warmup, epochs_per_step, target_seconds, scale, want_test, early_epochs=3):
    """Builds the workload `name`, makes its inputs resident, warms up, times `steps` steps and returns
    (contract fields + roofline of this config, the pieces the reporting-only legs need)."""
    from lightfm_amd import _native as N
    from lightfm_amd._lightfm_fast import CSRMatrix, make_opts
    from lightfm_amd.distributed import DistributedFit, MergePolicy
    from lightfm_amd.lightfm import LightFM, _Session
    from lightfm_amd.options import options
    args, rank, world, dist, log = env.args, env.rank, env.world, env.dist, env.log
    cfg = CONFIGS[name]
    scaling = args.scaling or cfg["default_scaling"]
    if world == 1:
        scaling = cfg["default_scaling"]
    policy = MergePolicy()
    if args.merge_mode:
        policy.mode = args.merge_mode
    if args.merge_k:
        policy.merge_k = args.merge_k
    if args.merge_max:
        policy.merge_max = args.merge_max
    policy.overlap = bool(args.merge_overlap)
    dev_name, cus, hbm = N.device_info(env.local_rank)

    t0 = time.time()
    key = (cfg["shape"] if isinstance(cfg["shape"], str) else name, scaling, scale, want_test)
    if key in env.cache and world == 1:
        train, test, global_n, n_users, n_items = env.cache[key]
        feats = build_features(name, n_items)
    else:
        train, feats, test, global_n, n_users, n_items = build_workload(name, rank, world, scaling, scale, want_test)
        if world == 1:
            env.cache.clear()  # one workload at a time in host memory
            env.cache[key] = (train, test, global_n, n_users, n_items)
    if world > 1 and scaling == "weak":
        # every rank generated its own shard: the merge schedule must be derived from ONE global count
        import torch
        t = torch.tensor([train.nnz], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        global_n = int(t[0])
    if args.emulate_shard > 1 and world == 1:
        from lightfm_amd.distributed import local_shard
        train, _ = local_shard(train, 0, args.emulate_shard, rebase=True)
        n_users, global_n, test = train.shape[0], train.nnz, None
    log("%s: %d interactions (%d x %d) ready in %.1fs" % (name, train.nnz, n_users, n_items, time.time() - t0))
    loss, d = cfg["loss"], cfg["d"]
    n_item_feat = feats.shape[1] if feats is not None else n_items

    # The epoch loop is the product's: lightfm_amd.distributed.DistributedFit (at N = 1 an epoch of it is exactly an
    # epoch of LightFM.fit_partial: device shuffle, seeds, one lfm_session_epoch, the on-device finite check; at
    # N > 1 segments + RCCL merges of the replicated tables, hot rows at their own cadence).  This rank's shard
    # carries LOCAL user ids, and only its own users' rows exist anywhere (host or device).
    model = LightFM(no_components=d, loss=loss, random_state=10 + rank, max_sampled=MAX_SAMPLED,
                    item_alpha=args.item_alpha, user_alpha=args.user_alpha)
    policy.sparse = not args.merge_dense
    fit = DistributedFit(model, train, rank, world, device=env.local_rank, dist=dist, policy=policy,
                         global_n=global_n, item_features=feats, local_ids=True)
    session = fit.session
    hot = fit.hot[0]
    rows = np.ascontiguousarray(train.row, dtype=np.int32)
    log("%s: setup done in %.1fs on %s (%d CUs)" % (name, time.time() - t0, dev_name, cus))

    n_local = train.nnz
    all_stats = []

    def epoch():
        try:
            all_stats.extend(fit.epoch())
        except ValueError:
            raise SystemExit("model diverged")

    def barrier():
        if world > 1:
            fit.barrier()
            dist.barrier()

    # warm-up: the first epoch ramps the concurrency up.  The epochs right after it are timed on their own
    # (`early_epochs`: the regime of a short fit -- what the quality leg and the CPU baseline run -- where nearly
    # every interaction still finds a violator and updates); their duration also calibrates epochs_per_step.
    eps = max(1, epochs_per_step)
    epoch()
    barrier()
    all_stats.clear()
    t1 = time.perf_counter()
    for _ in range(early_epochs):
        epoch()
    barrier()
    t_early = time.perf_counter() - t1
    t_epoch = t_early / early_epochs
    early_pos = float(sum(st.counters[0] for st in all_stats))
    early = {"epochs": "2..%d of the same run" % (1 + early_epochs), "seconds": t_early,
             "updates_per_interaction": float(sum(st.counters[2] for st in all_stats)) / max(1.0, early_pos)}
    if epochs_per_step <= 0:
        eps = int(min(64, max(1, math.ceil(target_seconds / max(1, steps) / max(t_epoch, 1e-4)))))
        if world > 1:
            import torch
            te = torch.tensor([eps], dtype=torch.int64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            eps = int(te[0])
    for _ in range(max(0, warmup * eps - 1 - early_epochs)):
        epoch()
    barrier()  # lfm_session_epoch / check_finite synchronise the session's stream before returning
    all_stats.clear()
    merges0, mbytes0 = fit.merges, fit.merge_bytes
    epoch0 = int(getattr(model, "_trained_interactions", 0)) // max(1, global_n)
    t_start = time.perf_counter()
    for _ in range(steps * eps):
        epoch()
    barrier()
    elapsed = time.perf_counter() - t_start
    stats = list(all_stats)
    local_pos = float(sum(s.counters[0] for s in stats))
    if world > 1:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        cnt = torch.tensor([local_pos], dtype=torch.float64)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total_pos = float(cnt[0])
        e = torch.tensor([early_pos], dtype=torch.float64)
        dist.all_reduce(e, op=dist.ReduceOp.SUM)
        early_pos = float(e[0])
        e = torch.tensor([early["seconds"]], dtype=torch.float64)
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        early["seconds"] = float(e[0])
    else:
        total_pos = local_pos
    early["value"] = early_pos / early["seconds"]
    early["unit"] = "interactions/s"

    # roofline of the dominant kernel (the epoch kernel of the loss), this rank
    kernel_s = sum(s.kernel_ms for s in stats) / 1e3
    counters = [sum(s.counters[i] for s in stats) for i in range(4)]
    pos_csr_lens = np.bincount(rows, minlength=n_users)
    lens = pos_csr_lens[rows]
    mean_probe = float(np.mean(8 + 4 * np.ceil(np.log2(lens + 1.0))))
    f_i = float(feats.nnz) / feats.shape[0] if feats is not None else 1.0
    n_epochs = steps * eps
    n_examples = float(n_local) * n_epochs
    alg = algorithmic_bytes(loss, counters, d, 1.0, f_i, mean_probe, n_examples,
                            mean_kos_pos=float(np.mean(np.minimum(10, lens))))
    launches = sum(int(s.launches) for s in stats)
    ng, used = int(stats[-1].tile_ng), int(stats[-1].kernel_used)
    reg = bool(args.item_alpha or args.user_alpha)
    if used == 1 and int(getattr(stats[-1], "tile_ahead", 0)):
        kernel_name = "fit_warp_tile_ahead_kernel<10, false>"
    elif used == 1:
        kernel_name = "fit_warp_tile_kernel<%d, %d, false, false, %s, %s>" % (
            64 // ng, {4: 4, 2: 2, 1: 1}[ng], "true" if ng == 4 and not (options.debug & 64) else "false",
            "true" if reg else "false")
    elif used == 2:
        kernel_name = "fit_feat_kernel<%d, %d, false, %s>" % (N.LOSS_IDS[loss], 1 if d <= 64 else 2,
                                                              "true" if reg else "false")  # <loss id, NC, TIMED, REG>
    else:
        kernel_name = "fit_%s_kernel (generic)" % loss.replace("-", "_")
    achieved = alg / kernel_s / 1e9
    traffic, traffic_source, traffic_extra = committed_traffic(name, kernel_name, counters[0] / max(1, launches))
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "kernel": kernel_name, "algorithmic_bytes_per_launch": alg / launches,
                "algorithmic_bytes_per_interaction": alg / max(1.0, counters[0]),
                "avg_launch_ms": kernel_s * 1e3 / launches, "launches_per_epoch": launches / n_epochs,
                "avg_launch_note": "HIP events around each epoch's launches on the session's stream / launches.  Consecutive "
                                   "full-size launches alternate between two streams and overlap by the earlier one's draining "
                                   "tail: rocprofv3's per-kernel durations include that wait; the union of their intervals / "
                                   "launches (profiles/*_kernel_stats.txt) is the comparable figure",
                "kernel_time_fraction_of_step": kernel_s / elapsed,
                "interactions_per_wavefront_pass": ng, "interactions_in_flight": int(stats[-1].in_flight),
                "draws_per_interaction": counters[1] / max(1.0, counters[0]),
                "updates_per_interaction": counters[2] / max(1.0, counters[0])}
    if traffic is not None:
        roofline["traffic_over_algorithmic"] = traffic / (alg / launches)
        roofline["traffic_profiled_run"] = traffic_extra

    # Second ceiling of the update-heavy configurations: every updated cell is published with one
    # global_atomic_add_f32 per table (W, G), and the chip executes a fixed ~320 G of them per second
    # whatever the table size or allocation (tools/membench.hip, profiles/r02_membench.txt: 10 G
    # 128-B line-ops/s = 1.28 TB/s of atomic payload; round 3: the rate is per DWORD, a 64-bit CAS or a
    # float64 add costs two -- profiles/r03_membench_atomics.txt).  Reported next to the HBM roofline.
    n_upd = counters[2] if loss != "logistic" else n_examples
    rows_upd = (1.0 + 2.0 * f_i) if loss != "logistic" else (1.0 + f_i)
    atomics = float(n_upd) * rows_upd * (d + 1) * 2.0
    roofline["atomic_unit"] = {"achieved": atomics / kernel_s / 1e9, "peak": ATOMIC_PEAK_GOPS, "unit": "G float atomics/s",
                               "frac": atomics / kernel_s / 1e9 / ATOMIC_PEAK_GOPS,
                               "atomics_per_interaction": atomics / max(1.0, counters[0]),
                               "peak_source": "measured on this chip by tools/membench.hip (profiles/r02_membench.txt)"}
    if options.feat_kernel == 2 and used == 2:  # profiling build of the row-stream kernel
        ph = np.sum([list(s.phase_cycles) for s in stats], axis=0).astype(np.float64)
        roofline["phase_cycles_per_interaction"] = dict(zip(
            ("sampling", "entry_lists", "rep_gather", "rep_reduce", "score", "update_gather", "update_math_publish",
             "tail"), [round(float(x) / max(1.0, counters[0]), 1) for x in ph]))
    if options.warp_kernel == 2 and used == 1:  # profiling build: per-phase shader cycles per wavefront pass
        ph = np.sum([list(s.phase_cycles) for s in stats], axis=0).astype(np.float64)
        passes = counters[0] / float(max(1, ng))
        roofline["phase_cycles_per_pass"] = dict(zip(
            ("head", "gather", "score", "lookup", "acc_loads", "update", "tail", "unused"),
            [round(float(x) / passes, 1) for x in ph]))
    session.close()

    par = ("1 GPU" if world == 1 else
           "%s scaling over %d GPUs: %s; item tables merged over RCCL (%s; %s), %.1f merges per epoch, %.1f MB "
           "exchanged per rank and merge"
           % (scaling, world,
              "one COO row-sharded by user" if scaling == "strong" else "every rank its own full-size row shard",
              policy.mode, "dense all-reduce, synchronous" if args.merge_dense else
              "all-reduce over the compacted union of the rows touched since the last merge%s%s"
              % (", overlapped with the next segment" if policy.overlap else ", synchronous",
                 "; %d hot rows merged every %d interactions" % (len(hot), world << 17) if len(hot) else ""),
              (fit.merges - merges0) / float(n_epochs),
              (fit.merge_bytes - mbytes0) / 1e6 / max(1, fit.merges - merges0)))
    result = {
        "value": total_pos / elapsed, "unit": "interactions/s", "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed * 1e3 / steps,
        "config": {"workload": cfg["label"] % (global_n if scaling == "strong" or world == 1 else n_local),
                   "name": name, "epochs_per_step": eps, "ms_per_epoch": elapsed * 1e3 / n_epochs,
                   "timed_epochs": "epochs %d..%d of one continuing training run" % (epoch0 + 1, epoch0 + n_epochs),
                   "parallelism": par, "device": dev_name},
        "roofline": roofline, "scaling": scaling, "early_epochs": early,
    }
    if scale != 1.0:
        result["config"]["scale"] = scale
    if args.item_alpha or args.user_alpha:
        result["config"]["item_alpha"], result["config"]["user_alpha"] = args.item_alpha, args.user_alpha
    pieces = dict(train=train, test=test, feats=feats, n_users=n_users, n_items=n_items, loss=loss, d=d, cfg=cfg)
    return result, pieces


def reporting_legs(name, env, pieces, want_quality):
    """quality (precision@10, 3 seeds), cpu_baseline (the reference on the host cores) and end_to_end_fit:
    reporting only, outside every timed region."""
    from lightfm_amd.lightfm import LightFM
    args, log = env.args, env.log
    train, test, feats = pieces["train"], pieces["test"], pieces["feats"]
    n_users, n_items, loss, d, cfg = pieces["n_users"], pieces["n_items"], pieces["loss"], pieces["d"], pieces["cfg"]
    quality, cpu, fit = None, None, None
    q_epochs = 3
    # the quality / CPU legs of c3 run on a row sub-sample (the reference needs ~10 us per
    # interaction there); c2 on the full COO
    q_train, q_test, q_note = train, test, "the full %d-interaction COO of this workload" % train.nnz
    if name == "c3":
        nu = n_users // 8

        q_train, q_test = head_users(train, nu, n_items), (head_users(test, nu, n_items) if test is not None else None)
        q_note = ("the first %d users' %d interactions of this workload (1/8 row sub-sample), full item-side "
                  "tables" % (nu, q_train.nnz))
    if want_quality and q_test is not None:
        q_seeds = (7, 8, 9)
        p = []
        for seed in q_seeds:
            m = LightFM(no_components=d, loss=loss, random_state=seed, max_sampled=MAX_SAMPLED)
            m.fit(q_train, item_features=feats, epochs=q_epochs)
            p.append(precision_at_10(m, q_train, q_test, feats))
        quality = {"epochs": q_epochs, "precision_at_10": float(np.mean(p)), "precision_at_10_seeds": p,
                   "seeds": list(q_seeds), "eval_users": int(len(np.unique(q_test.row))), "data": q_note,
                   "metric": "precision_at_k(k=10) of lightfm/evaluation.py:14-87 on a held-out 5 percent of "
                             "the same synthetic process; mean over %d seeds of %d-epoch fits (the reference: one fit, "
                             "seed 7, its 16-thread Hogwild)" % (len(q_seeds), q_epochs)}
    if not args.no_cpu_baseline:
        try:
            if cfg["shape"] == "ml-20m":
                cpu, p_ref = reference_leg(name, q_train, q_test if quality is not None else None, feats,
                                           q_epochs, log, q_note)
                if quality is not None and p_ref is not None:
                    quality["precision_at_10_ref"] = p_ref
                    quality["delta"] = quality["precision_at_10"] - p_ref
                if cpu:
                    ncpu = os.cpu_count() or 1
                    nu = max(1000, n_users // (16 if name == "c2" else 64))
                    sub = head_users(q_train, nu, n_items) if nu < q_train.shape[0] else q_train
                    cpu["thread_scaling"] = thread_scaling(name, sub, feats, log, sorted({1, min(16, ncpu), ncpu}))
                    cpu["thread_scaling_sample"] = ("epoch 2 of a fresh fit on the first %d users' %d interactions, "
                                                    "threads -> interactions/s" % (nu, sub.nnz))
            else:
                # C4 / C5 shards: a row sub-sample (1/50 of the users, their interactions, the
                # full item-side tables), SURVEY.md 8(d)
                nu = max(1000, n_users // 50)
                keep = train.row < nu
                sub = sp.coo_matrix((train.data[keep], (train.row[keep], train.col[keep])),
                                    shape=(nu, n_items), dtype=np.float32)
                cpu, _ = reference_leg(name, sub, None, feats, 3, log,
                                       "the first %d users' %d interactions of this shard (1/50 row sub-"
                                       "sample) over the full item-side tables" % (nu, sub.nnz))
        except Exception as e:  # the baseline is reporting only; never fail the bench on it
            log("cpu_baseline failed: %r" % (e,))
    if not args.no_fit and cfg["shape"] == "ml-20m":
        fit_epochs = 10
        m = LightFM(no_components=d, loss=loss, random_state=3, max_sampled=MAX_SAMPLED)
        t1 = time.perf_counter()
        m.fit(train, item_features=feats, epochs=fit_epochs)
        dt = time.perf_counter() - t1
        fit = {"value": train.nnz * fit_epochs / dt, "unit": "interactions/s", "epochs": fit_epochs,
               "seconds": dt, "what": "LightFM.fit(train, epochs=%d) through the public API: host coercion, "
               "uploads, device positives build, epochs, finite checks, download" % fit_epochs}
    return quality, cpu, fit


def main():
    args = parse_args()
    env = Env(args)
    rank, world = env.rank, env.world
    from lightfm_amd import _native as N
    from lightfm_amd.options import options
    # liblfm_hip.so (and with it /opt/rocm's HIP runtime, the one its kernels and librccl were built
    # for) is loaded BEFORE torch brings its own copy of the runtime into the process
    n_devices = N.device_count()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        env.dist = dist
    tuned = {k: getattr(args, k) for k in KNOBS if getattr(args, k) is not None}
    options.set(**tuned)
    if n_devices <= env.local_rank:
        raise SystemExit("no HIP device for local rank %d" % env.local_rank)

    name = args.config or "c2"
    cfg = CONFIGS[name]
    want_quality = world == 1 and not args.no_quality and cfg["shape"] == "ml-20m" and args.emulate_shard <= 1
    result, pieces = run_config(name, env, args.steps, args.warmup, args.epochs_per_step, 6.0, args.scale, want_quality)

    extras = []
    if world == 1 and args.config is None and not args.no_extra and not tuned:
        # the other BASELINE shapes, short legs timed the same way (contract: barrier + sync around K steps)
        plans = {"c3": dict(steps=3, warmup=1, target=2.5), "c4shard": dict(steps=3, warmup=1, target=1.5),
                 "c5shard": dict(steps=2, warmup=1, target=0.0, early=1)}
        for extra in [e for e in args.extra.split(",") if e in plans and e != name]:
            try:
                pl = plans[extra]
                r, pc = run_config(extra, env, pl["steps"], pl["warmup"], 0 if pl["target"] else 1, pl["target"], 1.0,
                                   False, early_epochs=pl.get("early", 3))
                leg = {"name": extra, "metric": "positive interactions/sec/epoch (%s, %s)" % (CONFIGS[extra]["loss"], extra),
                       "value": r["value"], "unit": r["unit"], "steps": r["steps"], "warmup": r["warmup"],
                       "ms_per_step": r["ms_per_step"], "config": r["config"], "roofline": r["roofline"],
                       "early_epochs": r["early_epochs"]}
                if not args.no_cpu_baseline:  # the reference on a bounded row sub-sample of THIS leg's workload
                    try:
                        leg["cpu_baseline"] = extra_cpu_leg(extra, env, pc)
                        if leg["cpu_baseline"]:
                            leg["speedup_vs_cpu_baseline"] = r["value"] / leg["cpu_baseline"]["value"]
                    except Exception as e:  # reporting only
                        env.log("cpu_baseline of %s failed: %r" % (extra, e))
                extras.append(leg)
            except BaseException as e:  # an extra leg never takes the contract line down
                env.log("extra config %s failed: %r" % (extra, e))
                extras.append({"name": extra, "error": repr(e)})
        env.cache.clear()

    quality, cpu, fit = None, None, None
    if rank == 0 and world == 1:
        quality, cpu, fit = reporting_legs(name, env, pieces, want_quality)

    if rank == 0:
        out = {
            "metric": "positive interactions/sec/epoch (WARP, ML-20M)" if name == "c2" else
                      "positive interactions/sec/epoch (%s, %s)" % (cfg["loss"], name),
            "value": result["value"], "unit": "interactions/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": result["ms_per_step"],
            "higher_is_better": True, "scaling": result["scaling"] if world > 1 else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": result["config"],
            "roofline": result["roofline"],
            "early_epochs": result["early_epochs"],
            "cpu_baseline": cpu,
            "quality": quality,
            "end_to_end_fit": fit,
        }
        if extras:
            out["extra_configs"] = extras
        if tuned:
            out["config"]["non_default_options"] = tuned
        if cpu:
            out["speedup_vs_cpu_baseline"] = result["value"] / cpu["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        env.dist.destroy_process_group()


if __name__ == "__main__":
    main()
