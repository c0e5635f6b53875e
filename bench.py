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

A "step" is ONE EPOCH of a FRESH fit (`--epochs-per-step` > 1 groups epochs): W warm-up epochs (the first ramps
the concurrency up), then exactly K timed epochs between barrier + synchronisation -- with the driver's
`--steps 20 --warmup 5` epochs 6..25 of a model that starts from its random initialisation, the regime a user's
`LightFM.fit` runs in (nearly every interaction still finds a violator and updates).  At N = 1 the measurement is
repeated on three fresh fits and `value` is the MEDIAN fit (`config.fresh_fits` lists all three); `config` also
carries `epochs_2_11` (the same fits' epochs 2..11) and `steady_state` (the last fit continued for a few seconds:
the regime of round 4's headline, fewer updates per interaction).  Every epoch is what LightFM.fit_partial does
per epoch -- the keyed on-device shuffle, the kernel seeds, one pass of the hot path over all interactions through
the C ABI (include/lfm_hip.h: lfm_session_epoch), the on-device finite check.  Inputs (weights, COO, positives
lookup) are resident in HBM before the timed region.

N > 1 (launched by torch.distributed.run, one rank per GPU).  c2 / c3 default to STRONG scaling:
ONE ML-20M-shaped COO sharded row-wise (by user) over the N ranks, user tables partitioned (a
rank allocates only its own users' rows), item tables replicated and merged by RCCL all-reduce
of their deltas at the cadence of lightfm_amd/distributed.py (merge_schedule), inside the timed
region; `--scaling weak` gives every rank its own full-size shard instead.  c4shard / c5shard
are per-GPU shards by definition (weak); `--item-tables owner` runs them with owner-sharded item tables
(HIP IPC peer mappings, no merges).  torch.distributed (gloo) is used only to hand the RCCL
unique id (or the IPC handles) to the ranks, for the barriers and for the max-over-ranks of the elapsed time.

The line stays below 8 KB (the driver keeps `config`, `roofline`, `cpu_baseline` and a 9 KB tail of stdout): numbers
are rounded to 5 significant digits, prose lives in profiles/README.md ("bench line").  Besides the contract fields:
`roofline` (dominant kernel, algorithmic bytes per SURVEY.md 8(d), HIP-event launch times, committed counter
traffic), `cpu_baseline` (the reference's own compiled Cython/OpenMP path on this box's host cores, N = 1 only) and,
inside `config`: `quality` (precision@10 of this backend and of the reference on the same data), `end_to_end_fit`
(LightFM.fit through the public API, uploads and downloads included) and `legs` -- short measurements of the
other BASELINE shapes (c3, c4shard, c5shard at its full per-GPU size), of the reference's DEFAULT width
(`c2_d10`: no_components = 10, rows padded to 12 floats on the device; `default_model`: LightFM()'s literal defaults, logistic
at no_components = 10, on +-1 labels; `c2_bpr`: BPR at c2's width) and of `predict_ranks` (all users x all items
of the ML-20M shape on the matrix cores), each with its own value, roofline and cpu baseline (--no-extra skips them).

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
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak, /opt/skills/guides/cdna_hip_programming.md (quick-reference table)
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
    # the reference's literal default model, LightFM(): logistic loss, no_components = 10 (LFM:191), on +-1 labels; and BPR at c2's width
    "default_model": dict(loss="logistic", d=10, shape="ml-20m", features=None, default_scaling="strong", labels="signed",
                          label="MovieLens-20M shape (138493 users x 26744 items x %d interactions, +-1 labels), loss=logistic, "
                                "no_components=10 (LightFM()'s defaults), identity features, adagrad"),
    "c2_bpr": dict(loss="bpr", d=64, shape="ml-20m", features=None, default_scaling="strong",
                   label="MovieLens-20M shape (138493 users x 26744 items x %d interactions), loss=bpr, no_components=64, "
                         "identity features, adagrad"),
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


def rnd(x, digits=5):
    """The line's numbers at 5 significant digits (it has to stay below 8 KB)."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        if x == 0.0:
            return 0.0
        return float("%.*g" % (digits, x))
    if isinstance(x, (np.floating,)):
        return rnd(float(x), digits)
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, dict):
        return {k: rnd(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [rnd(v, digits) for v in x]
    return x


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
    name = name if name in CONFIGS else "c2"
    nu = max(1000, n_users // frac)
    sub = head_users(train, nu, n_items)
    cpu, _ = reference_leg(name, sub, None, feats, 2, env.log,
                           "first %d users (%d interactions, 1/%d row sub-sample), full item side" % (nu, sub.nnz, frac))
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
           "sample": sample_note + "; native epoch call, mean of epochs 2..%d" % epochs}
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
        if e and kernel_name.split(" + ")[0] in e.get("kernel", ""):  # ("kernel + companion": the first one is the dominant kernel)
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
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--fits", type=int, default=3, help="N = 1: fresh fits measured; value = the median fit")
    ap.add_argument("--steady-seconds", type=float, default=4.0, help="N = 1: the last fit continued this long (config.steady_state)")
    ap.add_argument("--no-components", type=int, default=None, help="override the config's no_components")
    ap.add_argument("--item-tables", choices=("replicated", "owner"), default="replicated",
                    help="N > 1, identity WARP configs: owner-sharded item tables over HIP IPC instead of RCCL merges")
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="default: c2, followed at N = 1 by short legs of c3 / c4shard / c5shard (extra_configs)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs legs of the default run")
    ap.add_argument("--extra", default="ranks,c2_d10,default_model,c2_bpr,c3,c4shard,c5shard", help="which extra legs (comma separated)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default=None)
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the interactions (debug)")
    ap.add_argument("--emulate-shard", type=int, default=0,
                    help="debug, N = 1: run rank 0's row shard of a K-way strong-scaling split on this one GPU")
    ap.add_argument("--epochs-per-step", type=int, default=1, help="epochs of the fresh fit per step")
    ap.add_argument("--all-ranks-on-device", type=int, default=None,
                    help="debug: every rank uses this one GPU (owner-sharded item tables work across processes of one GPU; "
                         "RCCL refuses two ranks on one device)")
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
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0")) if args.all_ranks_on_device is None else int(args.all_ranks_on_device)
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus and self.world > 1:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world))
        self.dist = None
        self.cache = {}  # generated workloads shared between legs (c2 and c3 train on the same COO)

    def log(self, msg):
        if self.rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)


def kernel_label(loss, d, stats_last, reg, options, sharded=False):
    """Name of the epoch kernel a run's launches used (as rocprofv3 prints it)."""
    from lightfm_amd import _native as N
    ng, used = int(stats_last.tile_ng), int(stats_last.kernel_used)
    dp = (d + 3) // 4 * 4
    flags = int(getattr(stats_last, "plan_flags", 0))
    if used == 1 and flags & (256 | 512):  # the narrow lane-group kernels (csrc/logistic_tile.hip)
        return "fit_logistic_tile_kernel" if flags & 256 else "fit_bpr_tile_kernel<1>"
    if used == 1 and flags & 64:  # the narrow-model kernel: <candidates, USTORE, W and G of a row in one line, ... with the bias cells>
        rp = os.environ.get("LIGHTFM_AMD_ROW_PAIRS", "2") not in ("", "0")
        return "fit_warp_tile_narrow_kernel<10, %s, %s, %s>" % ("true" if int(getattr(stats_last, "user_store", 0)) else "false",
                                                                "true" if rp else "false", "true" if flags & 128 else "false")
    if used == 1 and int(getattr(stats_last, "tile_ahead", 0)):  # <candidates, owner-sharded item tables, user rows by plain stores>
        narrow = dp <= 16 and not sharded and os.environ.get("LIGHTFM_AMD_TILE_NARROW", "0") not in ("", "0")
        return "fit_warp_tile_ahead_kernel<10, %s, %s, %d>" % ("true" if sharded else "false",
                                                               "true" if int(getattr(stats_last, "user_store", 0)) else "false",
                                                               1 if narrow else 4)  # <candidates, SHARDED, USTORE, VEC>
    if used == 1:
        if flags & (1024 | 2048):  # fit_bpr / fit_logistic on the tile kernel's instantiations (csrc/warp_tile_bpr.hip: four floats of a row per lane)
            return "fit_warp_tile_kernel<%d, 4, false, false, %s, false, %d>" % (64 // ng, "true" if ng == 4 and not (options.debug & 64) else "false",
                                                                                 2 if flags & 1024 else 0)
        return "fit_warp_tile_kernel<%d, %d, false, false, %s, %s, 1>" % (
            64 // ng, {4: 4, 2: 2, 1: 1}[ng], "true" if ng == 4 and not (options.debug & 64) else "false",
            "true" if reg else "false")  # <lanes per row, floats per lane, TIMED, ADADELTA, DMA4, REG, loss id>
    if used == 2:
        hot = bool(int(getattr(stats_last, "plan_flags", 0)) & 32)  # the shared rows in LDS slices (csrc/hot_slices.hip)
        ada = bool(getattr(stats_last, "_adadelta", False))
        return "fit_feat_kernel<%d, %d, false, %s, %s, %s>%s" % (
            N.LOSS_IDS[loss], 1 if dp <= 64 else (2 if dp <= 128 else 4), "true" if reg else "false",
            "true" if hot else "false", "true" if ada else "false",
            " + hot_slice_kernel<8>" if hot else "")  # <loss id, NC, TIMED, REG, HOT, ADA>
    return "fit_%s_kernel (generic)" % loss.replace("-", "_")


def run_config(name, env, steps, warmup, epochs_per_step=1, fits=1, steady_seconds=0.0, scale=1.0, want_test=False,
               d_override=None):
    """Builds the workload `name`, makes its inputs resident and measures `fits` FRESH fits: `warmup` untimed steps, then
    exactly `steps` steps between barriers (a step = `epochs_per_step` epochs).  Returns (contract fields + roofline of
    the median fit, the pieces the reporting-only legs need)."""
    from lightfm_amd import _native as N
    from lightfm_amd.distributed import DistributedFit, MergePolicy
    from lightfm_amd.lightfm import LightFM
    from lightfm_amd.options import options
    args, rank, world, dist, log = env.args, env.rank, env.world, env.dist, env.log
    cfg = CONFIGS[name]
    scaling = args.scaling or cfg["default_scaling"]
    if world == 1:
        scaling = cfg["default_scaling"]
        if args.emulate_shard <= 1:
            pass
    else:
        fits = 1  # one communicator per process
    policy = MergePolicy()
    if args.merge_mode:
        policy.mode = args.merge_mode
    if args.merge_k:
        policy.merge_k = args.merge_k
    if args.merge_max:
        policy.merge_max = args.merge_max
    policy.overlap = bool(args.merge_overlap)
    policy.sparse = not args.merge_dense
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
    if cfg.get("labels") == "signed":  # logistic trains on both labels (PYX:751-755: y <= 0 is the label 0)
        sign = np.where(np.random.RandomState(3).rand(train.nnz) < 0.5, 1.0, -1.0).astype(np.float32)
        train = sp.coo_matrix((sign, (train.row, train.col)), shape=train.shape, dtype=np.float32)
    log("%s: %d interactions (%d x %d) ready in %.1fs" % (name, train.nnz, n_users, n_items, time.time() - t0))
    loss, d = cfg["loss"], (d_override or cfg["d"])
    n_local = train.nnz
    eps = max(1, epochs_per_step)
    rows = np.ascontiguousarray(train.row, dtype=np.int32)
    owner = args.item_tables == "owner" and world > 1 and feats is None and loss == "warp"

    def all_max(x):
        if world == 1:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def all_sum(x):
        if world == 1:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t[0])

    # The epoch loop is the product's: lightfm_amd.distributed.DistributedFit (at N = 1 an epoch of it is exactly an
    # epoch of LightFM.fit_partial: device shuffle, seeds, one lfm_session_epoch, the on-device finite check; at
    # N > 1 segments + RCCL merges of the replicated tables, hot rows at their own cadence).  This rank's shard
    # carries LOCAL user ids, and only its own users' rows exist anywhere (host or device).
    runs, steady, fit, model = [], None, None, None
    for f in range(max(1, fits)):
        if fit is not None:
            fit.close()
        model = LightFM(no_components=d, loss=loss, random_state=10 + rank + 100 * f, max_sampled=MAX_SAMPLED,
                        item_alpha=args.item_alpha, user_alpha=args.user_alpha)
        fit = DistributedFit(model, train, rank, world, device=env.local_rank, dist=dist, policy=policy,
                             global_n=global_n, item_features=feats, local_ids=True,
                             item_tables="owner" if owner else "replicated")
        if f == 0:
            log("%s: setup done in %.1fs on %s (%d CUs)" % (name, time.time() - t0, dev_name, cus))
        stats, walls = [], []

        def epoch():
            t1 = time.perf_counter()
            try:
                stats.extend(fit.epoch())  # (returns with the rank's stream drained)
            except ValueError:
                raise SystemExit("model diverged")
            walls.append(time.perf_counter() - t1)

        def barrier():
            if world > 1:
                fit.barrier()
                dist.barrier()

        for _ in range(warmup * eps):
            epoch()
        barrier()
        n_warm = len(stats)
        merges0, mbytes0 = fit.merges, fit.merge_bytes
        t_start = time.perf_counter()
        for _ in range(steps * eps):
            epoch()
        barrier()
        elapsed = all_max(time.perf_counter() - t_start)
        timed = stats[n_warm:]
        # (logistic visits every record; its first counter counts the label-1 records only)
        total_pos = all_sum(float(n_local * len(timed)) if loss == "logistic" else float(sum(st.counters[0] for st in timed)))
        run = dict(elapsed=elapsed, stats=timed, total_pos=total_pos, value=total_pos / elapsed,
                   merges=fit.merges - merges0, merge_bytes=fit.merge_bytes - mbytes0)
        if world == 1 and len(walls) >= 11:  # epochs 2..11 of this fresh fit (wall time per epoch, every epoch ends synchronised)
            run["epochs_2_11"] = 10.0 * n_local / sum(walls[1:11])
        runs.append(run)
        log("%s fit %d: %.4g interactions/s over epochs %d..%d" % (name, f, run["value"], warmup * eps + 1, (warmup + steps) * eps))
    # steady state: the last fit continued (round 4's headline regime: most interactions no longer update)
    if steady_seconds > 0 and world == 1:
        t1 = time.perf_counter()
        n_ep = 0
        while time.perf_counter() - t1 < 0.4 * steady_seconds:
            fit.epoch()
            n_ep += 1
        first = (warmup + steps) * eps + n_ep + 1
        t1 = time.perf_counter()
        ss, n_ss = [], 0
        while time.perf_counter() - t1 < 0.6 * steady_seconds:
            ss.extend(fit.epoch())
            n_ss += 1
        dt = time.perf_counter() - t1
        pos = float(sum(st.counters[0] for st in ss))
        steady = {"value": pos / dt, "epochs": "%d..%d" % (first, first + n_ss - 1), "ms_per_epoch": dt * 1e3 / max(1, n_ss),
                  "updates_per_interaction": float(sum(st.counters[2] for st in ss)) / max(1.0, pos),
                  "kernel_frac": algorithmic_frac(loss, ss, d, feats, rows, n_users, n_local * n_ss)}

    order = sorted(range(len(runs)), key=lambda q: runs[q]["value"])
    med = runs[order[len(order) // 2]]
    stats, elapsed, total_pos = med["stats"], med["elapsed"], med["total_pos"]

    # roofline of the dominant kernel (the epoch kernel of the loss), this rank, the median fit
    kernel_s = sum(st.kernel_ms for st in stats) / 1e3
    counters = [sum(st.counters[i] for st in stats) for i in range(4)]
    n_epochs = steps * eps
    alg = algorithmic_total(loss, counters, d, feats, rows, n_users, float(n_local) * n_epochs)
    visited = float(n_local) * n_epochs if loss == "logistic" else max(1.0, counters[0])  # (logistic: its first counter = the label-1 records)
    f_i = float(feats.nnz) / feats.shape[0] if feats is not None else 1.0
    launches = sum(int(st.launches) for st in stats)
    reg = bool(args.item_alpha or args.user_alpha)
    kernel_name = kernel_label(loss, d, stats[-1], reg, options, sharded=owner)
    achieved = alg / kernel_s / 1e9
    traffic, traffic_source, traffic_extra = committed_traffic(name if d_override is None else name + "_d%d" % d, kernel_name,
                                                               counters[0] / max(1, launches))
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "kernel": kernel_name, "algorithmic_bytes_per_launch": alg / launches,
                "algorithmic_bytes_per_interaction": alg / visited,
                "avg_launch_ms": kernel_s * 1e3 / launches, "launches_per_epoch": launches / n_epochs,
                "kernel_time_fraction_of_step": kernel_s / elapsed,
                "interactions_in_flight": int(stats[-1].in_flight),
                "draws_per_interaction": counters[1] / visited,
                "updates_per_interaction": counters[2] / visited}
    if traffic is not None:
        roofline["traffic_over_algorithmic"] = traffic / (alg / launches)
        if "traffic_over_algorithmic_profiled" in traffic_extra:
            roofline["traffic_over_algorithmic_profiled"] = traffic_extra["traffic_over_algorithmic_profiled"]
    # Second ceiling of the update-heavy configurations: every updated cell is published with one
    # global_atomic_add_f32 per table (W, G), and the chip executes a fixed ~320 G of them per second
    # (tools/membench.hip, profiles/r02_membench.txt; profiles/README.md "bench line").
    n_upd = counters[2] if loss != "logistic" else float(n_local) * n_epochs
    ustore = bool(int(getattr(stats[-1], "user_store", 0)))  # user rows written with plain stores (lfm_opts.user_store)
    rows_upd = ((0.0 if ustore else 1.0) + 2.0 * f_i) if loss != "logistic" else ((0.0 if ustore else 1.0) + f_i)
    hot_slices = bool(int(getattr(stats[-1], "plan_flags", 0)) & 32)
    if hot_slices:
        # the shared rows (fit.hot[0]: the same rule as the session's hot set) are accumulated in LDS slices and published once
        # per launch and slice replica (csrc/hot_slices.hip): what still goes through the float-atomic unit per interaction
        # is the rest of the rows
        n_hot = len(fit.hot[0])
        csc = feats.tocsc() if feats is not None else None
        hot_share = float(np.diff(csc.indptr)[fit.hot[0]].sum()) / max(1, csc.nnz) if csc is not None and n_hot else 0.0
        f_rest = f_i * (1.0 - hot_share)
        rows_upd = ((0.0 if ustore else 1.0) + 2.0 * f_rest) if loss != "logistic" else ((0.0 if ustore else 1.0) + f_rest)
        roofline["hot_slices"] = {"hot_rows": n_hot, "share_of_feature_entries": hot_share,
                                  "kernels": "fit_feat_kernel<..., HOT> writes one 256-B record + the user representation per "
                                             "interaction; hot_slice_kernel<8> applies them to LDS slices between launches"}
    atomics = float(n_upd) * rows_upd * (d + 1) * 2.0
    roofline["user_rows_by_plain_stores"] = ustore
    roofline["atomic_unit"] = {"achieved": atomics / kernel_s / 1e9, "peak": ATOMIC_PEAK_GOPS, "unit": "G float atomics/s",
                               "frac": atomics / kernel_s / 1e9 / ATOMIC_PEAK_GOPS,
                               "atomics_per_interaction": atomics / visited}
    if options.feat_kernel == 2 and int(stats[-1].kernel_used) == 2:  # profiling build of the row-stream kernel
        ph = np.sum([list(st.phase_cycles) for st in stats], axis=0).astype(np.float64)
        roofline["phase_cycles_per_interaction"] = dict(zip(
            ("sampling", "entry_lists", "rep_gather", "rep_reduce", "score", "update_gather", "update_math_publish",
             "tail"), [round(float(x) / max(1.0, counters[0]), 1) for x in ph]))
    if options.warp_kernel == 2 and int(stats[-1].kernel_used) == 1:  # profiling build: per-phase shader cycles per wavefront pass
        ph = np.sum([list(st.phase_cycles) for st in stats], axis=0).astype(np.float64)
        passes = counters[0] / float(max(1, int(stats[-1].tile_ng)))
        roofline["phase_cycles_per_pass"] = dict(zip(
            ("head", "gather", "score", "lookup", "acc_loads", "update", "tail", "unused"),
            [round(float(x) / passes, 1) for x in ph]))
    hot = fit.hot[0]
    fit.close()

    par = ("1 GPU" if world == 1 else
           "%s scaling over %d GPUs: %s; %s"
           % (scaling, world,
              "one COO row-sharded by user" if scaling == "strong" else "every rank its own full-size row shard",
              "item tables owner-sharded over HIP IPC peer mappings (no replicas, no merges)" if owner else
              "item tables merged over RCCL (%s; %s), %.1f merges per epoch, %.1f MB exchanged per rank and merge"
              % (policy.mode, "dense all-reduce, synchronous" if args.merge_dense else
                 "all-reduce over the compacted union of the rows touched since the last merge (all rows once a union "
                 "covered 90 %% of them)%s%s"
                 % (", overlapped with the next segment" if policy.overlap else ", synchronous",
                    "; %d hot rows merged every %d interactions" % (len(hot), world << 17) if len(hot) else ""),
                 med["merges"] / float(n_epochs), med["merge_bytes"] / 1e6 / max(1, med["merges"]))))
    first_timed = warmup * eps + 1
    result = {
        "value": total_pos / elapsed, "unit": "interactions/s", "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed * 1e3 / steps,
        "config": {"workload": cfg["label"] % (global_n if scaling == "strong" or world == 1 else n_local),
                   "name": name, "epochs_per_step": eps, "ms_per_epoch": elapsed * 1e3 / n_epochs,
                   "timed_epochs": "%d..%d of a fresh fit" % (first_timed, first_timed + n_epochs - 1),
                   "parallelism": par, "device": dev_name},
        "roofline": roofline, "scaling": scaling,
    }
    if d_override is not None:
        result["config"]["workload"] = result["config"]["workload"].replace("no_components=%d" % cfg["d"], "no_components=%d" % d)
    if len(runs) > 1:
        result["config"]["fresh_fits"] = [r["value"] for r in runs]
    e211 = [r["epochs_2_11"] for r in runs if "epochs_2_11" in r]
    if e211:
        result["config"]["epochs_2_11"] = float(np.median(e211))
    if steady:
        result["config"]["steady_state"] = steady
    if scale != 1.0:
        result["config"]["scale"] = scale
    if args.item_alpha or args.user_alpha:
        result["config"]["item_alpha"], result["config"]["user_alpha"] = args.item_alpha, args.user_alpha
    pieces = dict(train=train, test=test, feats=feats, n_users=n_users, n_items=n_items, loss=loss, d=d, cfg=cfg)
    return result, pieces


def algorithmic_total(loss, counters, d, feats, rows, n_users, n_examples):
    """SURVEY.md 8(d)'s bytes of the interactions a set of epochs processed."""
    lens = np.bincount(rows, minlength=n_users)[rows]
    mean_probe = float(np.mean(8 + 4 * np.ceil(np.log2(lens + 1.0))))
    f_i = float(feats.nnz) / feats.shape[0] if feats is not None else 1.0
    return algorithmic_bytes(loss, counters, d, 1.0, f_i, mean_probe, n_examples,
                             mean_kos_pos=float(np.mean(np.minimum(10, lens))))


def algorithmic_frac(loss, stats, d, feats, rows, n_users, n_examples):
    """Fraction of the HBM roofline of a list of epochs (algorithmic bytes / HIP-event kernel time / 8 TB/s)."""
    counters = [sum(st.counters[i] for st in stats) for i in range(4)]
    kernel_s = sum(st.kernel_ms for st in stats) / 1e3
    return algorithmic_total(loss, counters, d, feats, rows, n_users, n_examples) / kernel_s / 1e9 / HBM_PEAK_GBS


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
    q_train, q_test, q_note = train, test, "full COO (%d interactions)" % train.nnz
    if name == "c3":
        nu = n_users // 8
        q_train, q_test = head_users(train, nu, n_items), (head_users(test, nu, n_items) if test is not None else None)
        q_note = "first %d users (%d interactions, 1/8 row sub-sample), full item side" % (nu, q_train.nnz)
    if want_quality and q_test is not None:
        q_seeds = (7, 8, 9)
        p = []
        for seed in q_seeds:
            m = LightFM(no_components=d, loss=loss, random_state=seed, max_sampled=MAX_SAMPLED)
            m.fit(q_train, item_features=feats, epochs=q_epochs)
            p.append(precision_at_10(m, q_train, q_test, feats))
        quality = {"epochs": q_epochs, "precision_at_10": float(np.mean(p)), "seeds": p,
                   "eval_users": int(len(np.unique(q_test.row))), "data": q_note}
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
            else:
                # C4 / C5 shards: a row sub-sample (1/50 of the users, their interactions, the
                # full item-side tables), SURVEY.md 8(d)
                nu = max(1000, n_users // 50)
                sub = head_users(train, nu, n_items)
                cpu, _ = reference_leg(name, sub, None, feats, 3, log,
                                       "first %d users (%d interactions, 1/50 row sub-sample), full item side" % (nu, sub.nnz))
        except Exception as e:  # the baseline is reporting only; never fail the bench on it
            log("cpu_baseline failed: %r" % (e,))
    if not args.no_fit and cfg["shape"] == "ml-20m":
        fit_epochs = 10
        m = LightFM(no_components=d, loss=loss, random_state=3, max_sampled=MAX_SAMPLED)
        t1 = time.perf_counter()
        m.fit(train, item_features=feats, epochs=fit_epochs)
        dt = time.perf_counter() - t1
        fit = {"value": train.nnz * fit_epochs / dt, "epochs": fit_epochs, "seconds": dt}
    return quality, cpu, fit


def ranks_leg(env, pieces):
    """predict_ranks (PYX:1232-1323) of EVERY user with test interactions against all items of the
    ML-20M shape, on a model LightFM.fit trained -- the dense predict-all-items path on the matrix cores.  roofline: the
    2 * users * items * d flops of the score matrix against the dense fp32 matrix peak; cpu_baseline: the reference's
    own predict_ranks on a user sub-sample, 16 threads."""
    from lightfm_amd import _native as N
    from lightfm_amd.lightfm import LightFM
    train, test, n_users, n_items = pieces["train"], pieces["test"], pieces["n_users"], pieces["n_items"]
    if test is None:
        return None
    d = 64
    m = LightFM(no_components=d, loss="warp", random_state=1, max_sampled=MAX_SAMPLED)
    m.fit(train, epochs=2)
    test_csr, train_csr = test.tocsr().astype(np.float32), train.tocsr().astype(np.float32)
    users = int(np.count_nonzero(np.diff(test_csr.indptr)))
    walls, kms = [], []
    for _ in range(9):
        t1 = time.perf_counter()
        ranks = m.predict_rank(test_csr, train_interactions=train_csr, check_intersections=False)
        walls.append(time.perf_counter() - t1)
        kms.append(float(N.lib().lfm_last_kernel_ms()))
    pairs = float(users) * n_items
    k_ms, wall = float(np.median(kms[1:])), float(np.median(walls[1:]))
    leg = {"name": "predict_ranks", "metric": "user-item scores ranked/s (predict_ranks, ML-20M shape, no_components=64)",
           "value": pairs / (k_ms * 1e-3), "unit": "scores/s", "kernel_ms": k_ms, "call_ms": wall * 1e3,
           "call_ms_min_max": [min(walls[1:]) * 1e3, max(walls[1:]) * 1e3],
           "users": users, "items": n_items, "test_interactions": int(test_csr.nnz),
           "roofline": {"bound": "mfma", "achieved": 2.0 * d * pairs / (k_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": 2.0 * d * pairs / (k_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                        "frac_of_call": 2.0 * d * pairs / wall / 1e12 / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                        "kernel": "ranks_mfma3_kernel<.., bf16 pipe> + piece table / test_scores / sortedness passes (HIP events around the call's kernels)",
                        "products": "2 d flops per pair, priced against the fp32 matrix peak (the arithmetic the ranks are exact in); they RUN as three "
                                    "v_mfma_f32_32x32x16_bf16 per 16 components on two-way split operands (6 d flops per pair on a 2.5 PFLOP/s pipe) so that "
                                    "they co-execute with the VALU search that bounds the kernel"}}
    if not env.args.no_cpu_baseline:
        try:
            from oracle import oracle
            if oracle.ref_available("fast"):
                threads = min(16, os.cpu_count() or 1)
                eye_i = sp.identity(n_items, dtype=np.float32, format="csr")

                def reference_ranks(kind, nu):
                    """The reference's predict_ranks over the first nu users: (ranks, seconds, users with test interactions)."""
                    ref = oracle.ref_module(kind)
                    sub_test, sub_train = test_csr[:nu].tocsr(), train_csr[:nu].tocsr()
                    fl = ref.FastLightFM(m.item_embeddings, m.item_embedding_gradients, m.item_embedding_momentum, m.item_biases,
                                         m.item_bias_gradients, m.item_bias_momentum,
                                         *[np.ascontiguousarray(getattr(m, n_)[:nu]) for n_ in (
                                             "user_embeddings", "user_embedding_gradients", "user_embedding_momentum", "user_biases",
                                             "user_bias_gradients", "user_bias_momentum")],
                                         d, 0, m.learning_rate, m.rho, m.epsilon, m.max_sampled)
                    out = np.zeros(sub_test.nnz, np.float32)
                    t1 = time.perf_counter()
                    ref.predict_ranks(ref.CSRMatrix(eye_i), ref.CSRMatrix(sp.identity(nu, dtype=np.float32, format="csr")),
                                      ref.CSRMatrix(sub_test), ref.CSRMatrix(sub_train), out, fl, threads)
                    return out, time.perf_counter() - t1, int(np.count_nonzero(np.diff(sub_test.indptr)))

                # timed: the default-flag build users get (-ffast-math: its dot products may round differently);
                # compared: the strict build (the parity oracle) on a smaller sample -- ranks must be identical
                out, dt, scored_users = reference_ranks("fast", 1500)
                leg["cpu_baseline"] = {"value": float(scored_users) * n_items / dt, "unit": "scores/s", "cores": threads,
                                       "kind": "reference", "sample": "first 1500 users (%d with test interactions)" % scored_users,
                                       "ranks_equal_to_fast_build": float(np.mean(out == ranks.data[:len(out)]))}
                if oracle.ref_available("strict"):
                    out, _, _ = reference_ranks("strict", 400)
                    leg["cpu_baseline"]["ranks_identical_to_strict_build"] = bool(np.array_equal(out, ranks.data[:len(out)]))
                leg["speedup_vs_cpu_baseline"] = leg["value"] / leg["cpu_baseline"]["value"]
        except Exception as e:  # reporting only
            env.log("predict_ranks cpu_baseline failed: %r" % (e,))
    return leg


def compact_leg(r, cpu=None):
    """An extra leg's measurement in a few hundred bytes."""
    ro = r["roofline"]
    out = {"value": r["value"], "unit": r["unit"], "steps": r["steps"], "warmup": r["warmup"], "ms_per_step": r["ms_per_step"],
           "workload": r["config"]["workload"], "timed_epochs": r["config"]["timed_epochs"],
           "roofline": {k: ro[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms",
                                           "algorithmic_bytes_per_interaction", "updates_per_interaction", "draws_per_interaction",
                                           "traffic_over_algorithmic") if k in ro}}
    out["roofline"]["atomic_unit_frac"] = ro["atomic_unit"]["frac"]
    if "steady_state" in r["config"]:
        out["steady_state"] = r["config"]["steady_state"]
    if cpu:
        out["cpu_baseline"] = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample") if k in cpu}
        out["speedup_vs_cpu_baseline"] = r["value"] / cpu["value"]
    return out


def mini_leg(r):
    """... and in a few dozen: rate, roofline fraction, kernel."""
    ro = r["roofline"]
    return {"value": r["value"], "unit": r["unit"], "workload": r["config"]["workload"].split("), ", 1)[-1], "timed_epochs": r["config"]["timed_epochs"],
            "roofline_frac": ro["frac"], "kernel": ro["kernel"], "atomic_unit_frac": ro["atomic_unit"]["frac"]}


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
        N.preload_comm()  # the RCCL of OUR HIP runtime, before torch's bundled one is in the process
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
    want_ranks = world == 1 and args.config is None and not args.no_extra and not tuned
    result, pieces = run_config(name, env, args.steps, args.warmup, args.epochs_per_step, fits=args.fits,
                                steady_seconds=args.steady_seconds, scale=args.scale, want_test=want_quality or want_ranks,
                                d_override=args.no_components)

    legs = {}
    if world == 1 and args.config is None and not args.no_extra and not tuned:
        # reporting legs that share c2's COO first (the generated workload is cached)
        wanted = [e for e in args.extra.split(",") if e]
        if "ranks" in wanted:
            try:
                leg = ranks_leg(env, pieces)
                if leg:
                    legs["predict_ranks"] = leg
            except BaseException as e:
                env.log("predict_ranks leg failed: %r" % (e,))
                legs["predict_ranks"] = {"error": repr(e)}
        # the other BASELINE shapes and the reference's default width, short legs timed the same way
        plans = {"c2_d10": dict(cfg="c2", steps=5, warmup=2, d=10), "default_model": dict(cfg="default_model", steps=5, warmup=2, mini=True),
                 "c2_bpr": dict(cfg="c2_bpr", steps=5, warmup=2, mini=True), "c3": dict(cfg="c3", steps=5, warmup=2, steady=2.0),
                 "c4shard": dict(cfg="c4shard", steps=5, warmup=2, steady=1.5), "c5shard": dict(cfg="c5shard", steps=2, warmup=1)}
        for extra in [e for e in wanted if e in plans and e != name]:
            try:
                pl = plans[extra]
                r, pc = run_config(pl["cfg"], env, pl["steps"], pl["warmup"], 1, fits=1, steady_seconds=pl.get("steady", 0.0),
                                   d_override=pl.get("d"), want_test=(CONFIGS[pl["cfg"]]["shape"] == "ml-20m"))
                cpu = None
                if not args.no_cpu_baseline and extra != "c2_d10" and not pl.get("mini"):  # the reference on a bounded row sub-sample of THIS leg's workload
                    try:
                        cpu = extra_cpu_leg(extra, env, pc)
                    except Exception as e:  # reporting only
                        env.log("cpu_baseline of %s failed: %r" % (extra, e))
                legs[extra] = mini_leg(r) if pl.get("mini") else compact_leg(r, cpu)
            except BaseException as e:  # an extra leg never takes the contract line down
                env.log("extra config %s failed: %r" % (extra, e))
                legs[extra] = {"error": repr(e)}
        env.cache.clear()

    quality, cpu, fit = None, None, None
    if rank == 0 and world == 1:
        if name != "c2" or "c2" not in str(env.cache.keys()):
            pass
        quality, cpu, fit = reporting_legs(name, env, pieces, want_quality)

    if rank == 0:
        config = result["config"]
        if quality:
            config["quality"] = quality
        if fit:
            config["end_to_end_fit"] = fit
        if legs:
            config["legs"] = legs
        if tuned:
            config["non_default_options"] = tuned
        if cpu:
            config["speedup_vs_cpu_baseline"] = result["value"] / cpu["value"]
        out = {
            "metric": "positive interactions/sec/epoch (WARP, ML-20M)" if name == "c2" else
                      "positive interactions/sec/epoch (%s, %s)" % (cfg["loss"], name),
            "value": result["value"], "unit": "interactions/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": result["ms_per_step"],
            "higher_is_better": True, "scaling": result["scaling"] if world > 1 else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": config,
            "roofline": result["roofline"],
            "cpu_baseline": cpu,
        }
        line = json.dumps(rnd(out), separators=(",", ":"))
        if len(line) > 8000:
            env.log("bench line is %d bytes (> 8000): the driver's stdout tail may cut it" % len(line))
        print(line, flush=True)
    if world > 1:
        env.dist.destroy_process_group()


if __name__ == "__main__":
    main()
