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
    feats = None
    if cfg["features"] == "tags":
        feats = synthetic.tag_item_features(n_items)
    elif cfg["features"] == "hashed":
        feats = synthetic.hashed_item_features(n_items)
    return train, feats, test, global_n, train.shape[0], n_items


def precision_at_10(model, train, test, item_features):
    """The reference's precision_at_k (lightfm/evaluation.py:14-87), mean over ALL users with test
    interactions (the device ranks kernel scores every user x item pair in tens of milliseconds)."""
    from lightfm_amd.evaluation import precision_at_k
    return float(precision_at_k(model, test.tocsr(), train_interactions=train.tocsr(), k=10,
                                item_features=item_features).mean())


def reference_leg(cfg_name, train, test, feats, epochs, log, sample_note):
    """The reference's compiled Cython/OpenMP path (oracle/_ref/fast) on this box's host cores:
    throughput of its native epoch call (and with its per-epoch host prologue), and -- when a
    test set is given -- precision@10 of the model it trained."""
    from oracle import oracle
    from oracle.ref_model import RefLightFM
    if not oracle.ref_available("fast"):
        return None, None
    cfg = CONFIGS[cfg_name]
    ncpu = os.cpu_count() or 1
    threads = min(16, ncpu)  # the reference's Hogwild stops scaling there (measured: 16 beats 64 and 256)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
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
    for knob in ("update_mode", "first_batch", "launches_per_epoch", "max_waves", "warp_kernel", "feat_kernel",
                 "debug", "ramp_k", "shared_cap"):
        ap.add_argument("--" + knob.replace("_", "-"), type=int, default=None, help="backend option (tuning)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    cfg = CONFIGS[args.config]
    scaling = args.scaling or cfg["default_scaling"]
    if world == 1:
        scaling = cfg["default_scaling"]

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    from lightfm_amd import _native as N
    from lightfm_amd._lightfm_fast import CSRMatrix, FastLightFM, make_opts
    from lightfm_amd.distributed import MergePolicy, merge_schedule, segment_positions
    from lightfm_amd.lightfm import LightFM, _Session
    from lightfm_amd.options import options
    # liblfm_hip.so (and with it /opt/rocm's HIP runtime, the one its kernels and librccl were built
    # for) is loaded BEFORE torch brings its own copy of the runtime into the process
    n_devices = N.device_count()

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    tuned = {k: getattr(args, k) for k in ("update_mode", "first_batch", "launches_per_epoch", "max_waves",
                                           "warp_kernel", "feat_kernel", "debug", "ramp_k", "shared_cap")
             if getattr(args, k) is not None}
    options.set(**tuned)
    policy = MergePolicy()
    if args.merge_mode:
        policy.mode = args.merge_mode
    if args.merge_k:
        policy.merge_k = args.merge_k
    if args.merge_max:
        policy.merge_max = args.merge_max
    if n_devices <= local_rank:
        raise SystemExit("no HIP device for local rank %d" % local_rank)
    dev_name, cus, hbm = N.device_info(local_rank)

    t0 = time.time()
    want_quality = world == 1 and not args.no_quality and cfg["shape"] == "ml-20m"
    train, feats, test, global_n, n_users, n_items = build_workload(args.config, rank, world, scaling, args.scale,
                                                                     want_quality)
    if world > 1 and scaling == "weak":
        # every rank generated its own shard: the merge schedule must be derived from ONE global count
        import torch
        t = torch.tensor([train.nnz], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        global_n = int(t[0])
    if args.emulate_shard > 1 and world == 1:
        from lightfm_amd.distributed import local_shard
        train, _ = local_shard(train, 0, args.emulate_shard, rebase=True)
        n_users, global_n, want_quality = train.shape[0], train.nnz, False
    log("generated %d interactions (%d x %d) in %.1fs" % (train.nnz, n_users, n_items, time.time() - t0))
    loss, d = cfg["loss"], cfg["d"]
    n_item_feat = feats.shape[1] if feats is not None else n_items

    model = LightFM(no_components=d, loss=loss, random_state=10 + rank, max_sampled=MAX_SAMPLED)
    model._initialize(d, n_item_feat, n_users)
    if world > 1:  # replicated item tables start identical on every rank
        import torch
        dist.broadcast(torch.from_numpy(model.item_embeddings), src=0)
    item_f = feats if feats is not None else sp.identity(n_items, dtype=np.float32, format="csr")
    user_f = sp.identity(n_users, dtype=np.float32, format="csr")
    fl = model._get_lightfm_data()
    session = _Session(fl, CSRMatrix(item_f), CSRMatrix(user_f), device=local_rank)
    rows = np.ascontiguousarray(train.row, dtype=np.int32)
    cols = np.ascontiguousarray(train.col, dtype=np.int32)
    vals = np.ascontiguousarray(train.data, dtype=np.float32)
    session.set_interactions(None, rows, cols, vals, vals)
    session.build_positives(n_users, n_items)
    if world > 1:
        uid = C.create_string_buffer(N.UNIQUE_ID_BYTES)
        if rank == 0:
            N.check(N.lib().lfm_comm_unique_id(uid))
        import torch
        t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).clone()
        dist.broadcast(t, src=0)
        session.comm_init(C.create_string_buffer(bytes(t.numpy().tobytes()), N.UNIQUE_ID_BYTES), rank, world)
    log("setup done in %.1fs on %s (%d CUs)" % (time.time() - t0, dev_name, cus))

    n_local = train.nnz
    state = {"history": 0, "merges": 0}
    all_stats = []

    def epoch():
        """What LightFM.fit_partial / DistributedFit.run do per epoch."""
        keys = model.random_state.randint(0, np.iinfo(np.int32).max, size=624)
        session.device_shuffle(int(keys[0]), int(keys[1]))
        seeds = np.ascontiguousarray(model.random_state.randint(
            0, np.iinfo(np.int32).max, size=1).astype(np.uint32))
        if world == 1:
            opts, _ = make_opts()
            opts.history = state["history"]
            session.epoch(loss, args.item_alpha, args.user_alpha, 5, 10, seeds, opts)
            all_stats.append(opts)
        else:
            pos = segment_positions(merge_schedule(state["history"], global_n, world, policy), n_local)
            for j in range(len(pos) - 1):
                opts, _ = make_opts()
                opts.history = (state["history"] + int(round(global_n * pos[j] / max(1, n_local)))) // world
                opts.pos_begin, opts.pos_end = int(pos[j]), int(pos[j + 1])
                if pos[j + 1] > pos[j]:
                    session.epoch(loss, args.item_alpha, args.user_alpha, 5, 10, seeds, opts)
                    all_stats.append(opts)
                session.comm_merge(1, policy.mode_id())
                state["merges"] += 1
        state["history"] += global_n if world > 1 else n_local
        bad = not session.check_finite()
        if world > 1:
            bad = session.comm_any(bad)
        if bad:
            raise SystemExit("model diverged")

    def barrier():
        if world > 1:
            session.comm_barrier()
            dist.barrier()

    # warm-up: the first epoch ramps the concurrency up; its duration calibrates epochs_per_step
    eps = max(1, args.epochs_per_step)
    epoch()
    t1 = time.perf_counter()
    epoch()
    t_epoch = time.perf_counter() - t1
    if args.epochs_per_step <= 0:
        eps = int(min(64, max(1, math.ceil(6.0 / max(1, args.steps) / max(t_epoch, 1e-4)))))
        if world > 1:
            import torch
            te = torch.tensor([eps], dtype=torch.int64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            eps = int(te[0])
    for _ in range(max(0, args.warmup * eps - 2)):
        epoch()
    barrier()  # lfm_session_epoch / check_finite synchronise the session's stream before returning
    all_stats.clear()
    merges0 = state["merges"]
    t_start = time.perf_counter()
    for _ in range(args.steps * eps):
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
    else:
        total_pos = local_pos

    # roofline of the dominant kernel (the epoch kernel of the loss), this rank
    kernel_s = sum(s.kernel_ms for s in stats) / 1e3
    counters = [sum(s.counters[i] for s in stats) for i in range(4)]
    pos_csr_lens = np.bincount(rows, minlength=n_users)
    lens = pos_csr_lens[rows]
    mean_probe = float(np.mean(8 + 4 * np.ceil(np.log2(lens + 1.0))))
    f_i = float(feats.nnz) / feats.shape[0] if feats is not None else 1.0
    n_examples = float(n_local) * (args.steps * eps)
    alg = algorithmic_bytes(loss, counters, d, 1.0, f_i, mean_probe, n_examples,
                            mean_kos_pos=float(np.mean(np.minimum(10, lens))))
    launches = sum(int(s.launches) for s in stats)
    ng, used = int(stats[-1].tile_ng), int(stats[-1].kernel_used)
    if used == 1:
        kernel_name = "fit_warp_tile_kernel<%d, %d, false, false>" % (64 // ng, {4: 4, 2: 2, 1: 1}[ng])
    elif used == 2:
        kernel_name = "fit_feat_kernel (%s)" % loss
    else:
        kernel_name = "fit_%s_kernel (generic)" % loss.replace("-", "_")
    achieved = alg / kernel_s / 1e9
    n_epochs = args.steps * eps
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "traffic_note": "PMC traffic is not measurable inside this run; see profiles/README.md for "
                                "the rocprofv3 --pmc summary of the same command",
                "kernel": kernel_name, "algorithmic_bytes_per_launch": alg / launches,
                "algorithmic_bytes_per_interaction": alg / max(1.0, counters[0]),
                "avg_launch_ms": kernel_s * 1e3 / launches, "launches_per_epoch": launches / n_epochs,
                "kernel_time_fraction_of_step": kernel_s / elapsed,
                "interactions_per_wavefront_pass": ng, "interactions_in_flight": int(stats[-1].in_flight),
                "draws_per_interaction": counters[1] / max(1.0, counters[0]),
                "updates_per_interaction": counters[2] / max(1.0, counters[0])}

    # Second ceiling of the update-heavy configurations: every updated cell is published with one
    # global_atomic_add_f32 per table (W, G), and the chip executes a fixed ~320 G of them per second
    # whatever the table size or allocation (tools/membench.hip, profiles/r02_membench.txt: 10 G
    # 128-B line-ops/s = 1.28 TB/s of atomic payload).  Reported next to the HBM roofline.
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

    quality, cpu, fit = None, None, None
    if rank == 0 and world == 1:
        session.close()
        q_epochs = 3
        # the quality / CPU legs of c3 run on a row sub-sample (the reference needs ~10 us per
        # interaction there); c2 on the full COO
        q_train, q_test, q_note = train, test, "the full %d-interaction COO of this workload" % train.nnz
        if args.config == "c3":
            nu = n_users // 8
            def head(coo):
                keep = coo.row < nu
                return sp.coo_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=(nu, n_items),
                                     dtype=np.float32)
            q_train, q_test = head(train), (head(test) if test is not None else None)
            q_note = ("the first %d users' %d interactions of this workload (1/8 row sub-sample), full item-side "
                      "tables" % (nu, q_train.nnz))
        if want_quality:
            m = LightFM(no_components=d, loss=loss, random_state=7, max_sampled=MAX_SAMPLED)
            m.fit(q_train, item_features=feats, epochs=q_epochs)
            quality = {"epochs": q_epochs, "precision_at_10": precision_at_10(m, q_train, q_test, feats),
                       "eval_users": int(len(np.unique(q_test.row))), "data": q_note,
                       "metric": "precision_at_k(k=10) of lightfm/evaluation.py:14-87 on a held-out 5 percent of "
                                 "the same synthetic process; both backends fit %d epochs from the same seed" % q_epochs}
        if not args.no_cpu_baseline:
            try:
                if cfg["shape"] == "ml-20m":
                    cpu, p_ref = reference_leg(args.config, q_train, q_test if want_quality else None, feats,
                                               q_epochs, log, q_note)
                    if quality is not None and p_ref is not None:
                        quality["precision_at_10_ref"] = p_ref
                        quality["delta"] = quality["precision_at_10"] - p_ref
                else:
                    # C4 / C5 shards: a row sub-sample (1/50 of the users, their interactions, the
                    # full item-side tables), SURVEY.md 8(d)
                    nu = max(1000, n_users // 50)
                    keep = train.row < nu
                    sub = sp.coo_matrix((train.data[keep], (train.row[keep], train.col[keep])),
                                        shape=(nu, n_items), dtype=np.float32)
                    cpu, _ = reference_leg(args.config, sub, None, feats, 3, log,
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
    else:
        session.close()

    if rank == 0:
        value = total_pos / elapsed
        par = ("1 GPU" if world == 1 else
               "%s scaling over %d GPUs: %s; item tables merged by RCCL all-reduce (%s), %.1f merges per epoch"
               % (scaling, world,
                  "one COO row-sharded by user" if scaling == "strong" else "every rank its own full-size row shard",
                  policy.mode, (state["merges"] - merges0) / float(n_epochs)))
        out = {
            "metric": "positive interactions/sec/epoch (WARP, ML-20M)" if args.config == "c2" else
                      "positive interactions/sec/epoch (%s, %s)" % (loss, args.config),
            "value": value, "unit": "interactions/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": scaling if world > 1 else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["label"] % (global_n if scaling == "strong" or world == 1 else n_local),
                       "name": args.config, "epochs_per_step": eps, "ms_per_epoch": elapsed * 1e3 / n_epochs,
                       "parallelism": par, "device": dev_name},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "quality": quality,
            "end_to_end_fit": fit,
        }
        if tuned:
            out["config"]["non_default_options"] = tuned
        if args.item_alpha or args.user_alpha:
            out["config"]["item_alpha"], out["config"]["user_alpha"] = args.item_alpha, args.user_alpha
        if cpu:
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
