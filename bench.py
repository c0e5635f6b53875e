#!/usr/bin/env python
"""bench.py -- positive interactions/sec/epoch of the WARP epoch kernel on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1]): MovieLens-20M SHAPE (138,493 users x 26,744
items x 20,000,263 interactions; synthetic, no dataset is reachable offline),
loss='warp', no_components=64, identity features, adagrad, lr 0.05,
max_sampled=10, parallel (Hogwild) mode of the HIP backend with its defaults (atomic
publication, concurrency ramped with the training history: step e is epoch e of ONE training
run, so the ramp happens in the first warm-up epoch and the timed epochs run with the chip full).

A "step" is ONE EPOCH = one pass of the hot path over all interactions of the
rank's shard (every positive visited once, negatives sampled, Adagrad updates
applied), driven through the C ABI (include/lfm_hip.h: lfm_session_epoch) plus the
on-device finite check of LightFM.fit_partial.  Inputs (weights, COO, positives
CSR and one pre-shuffled index list per step) are resident in HBM before the timed
region starts.

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling -- every
rank owns its own 138,493 users and 20,000,263 interactions (a row shard of an
N x larger interaction matrix over the same 26,744 items); user-side tables are
partitioned and never communicated, the item-side tables are merged after every
epoch by an RCCL all-reduce of their deltas, inside the timed step.
torch.distributed (gloo) is used only to hand the RCCL unique id to the ranks, for
the barriers and for the max-over-ranks of the elapsed time.

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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md:35
D = 64
MAX_SAMPLED = 10


def algorithmic_bytes(n_pos, draws, updates, probes, d, f_u, f_i, mean_probe_bytes):
    """B = H + R(f_u) + (1+S) R(f_i) + U [Up(f_u) + 2 Up(f_i)] + V P."""
    H = 20.0
    R = lambda f: 8 + 8 * f + 4 * f * (d + 1)
    Up = lambda f: 16 * f * (d + 1)
    return (n_pos * (H + R(f_u) + R(f_i)) + draws * R(f_i)
            + updates * (Up(f_u) + 2 * Up(f_i)) + probes * mean_probe_bytes)


def cpu_baseline(train, log):
    """The reference's own compiled Cython/OpenMP path (oracle/_ref/fast) on this box's
    host cores, on a bounded sample of the same workload."""
    try:
        from oracle import oracle
        from oracle.ref_model import RefLightFM
        from lightfm_amd import synthetic
        if not oracle.ref_available("fast"):
            return None
        n_sample = 2_000_000
        rng = np.random.RandomState(0)
        idx = np.sort(rng.choice(train.nnz, size=min(n_sample, train.nnz), replace=False))
        import scipy.sparse as sp
        sample = sp.coo_matrix((train.data[idx], (train.row[idx], train.col[idx])),
                               shape=train.shape, dtype=np.float32)
        best = None
        ncpu = os.cpu_count() or 1
        for threads in sorted({ncpu, min(ncpu, 64), min(ncpu, 16)}, reverse=True):
            m = RefLightFM(no_components=D, loss="warp", random_state=10, max_sampled=MAX_SAMPLED)
            m.fit_partial(sample, epochs=1, num_threads=threads)  # warm-up epoch (page faults)
            t0 = time.time()
            m.fit_partial(sample, epochs=2, num_threads=threads)
            dt = (time.time() - t0) / 2
            rate = sample.nnz / dt
            log("cpu_baseline threads=%d: %.3g interactions/s" % (threads, rate))
            if best is None or rate > best[0]:
                best = (rate, threads)
        return {"value": best[0], "unit": "interactions/s", "cores": best[1], "kind": "reference",
                "sample": "%d-interaction random sub-sample of the same ML-20M-shaped COO over the "
                          "full-size tables, reference v1.17 Cython/OpenMP build (-O2 -ffast-math "
                          "-march=x86-64-v3 -fopenmp), epochs 2-3 after one warm-up epoch, best of "
                          "thread counts {all, 64, 16}; includes the reference's per-epoch host "
                          "prologue (tocsr + shuffle)" % sample.nnz,
                "host_cpus": ncpu}
    except Exception as e:  # the baseline is reporting only; never fail the bench on it
        log("cpu_baseline failed: %r" % (e,))
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the 20M interactions (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--debug-zipf", type=float, default=None, help="experiment: item popularity exponent")
    ap.add_argument("--debug-no-shuffle", action="store_true", help="experiment: identity shuffle")
    ap.add_argument("--debug-empty-positives", action="store_true", help="experiment: no in_positives probes")
    for knob in ("update_mode", "first_batch", "launches_per_epoch", "max_waves", "warp_kernel", "debug"):
        ap.add_argument("--" + knob.replace("_", "-"), type=int, default=None, help="backend option (tuning)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    from lightfm_amd import _native as N
    from lightfm_amd import synthetic
    from lightfm_amd._lightfm_fast import CSRMatrix
    from lightfm_amd.lightfm import LightFM, _Session
    import scipy.sparse as sp

    from lightfm_amd.options import options
    tuned = {k: getattr(args, k) for k in ("update_mode", "first_batch",
                                           "launches_per_epoch", "max_waves", "warp_kernel", "debug")
             if getattr(args, k) is not None}
    options.set(**tuned)
    if N.device_count() <= local_rank:
        raise SystemExit("no HIP device for local rank %d" % local_rank)
    dev_name, cus, hbm = N.device_info(local_rank)

    t0 = time.time()
    n_users, n_items, nnz = synthetic.SHAPES["ml-20m"]
    extra = {} if args.debug_zipf is None else {"zipf": args.debug_zipf}
    train = synthetic.make_interactions(n_users, n_items, int(nnz * args.scale), seed=42 + rank, **extra)
    log("generated %d interactions in %.1fs" % (train.nnz, time.time() - t0))

    model = LightFM(no_components=D, loss="warp", random_state=10 + rank, max_sampled=MAX_SAMPLED)
    model._initialize(D, n_items, n_users)
    if world > 1:  # replicated item tables start identical on every rank
        import torch
        for name in ("item_embeddings",):
            t = torch.from_numpy(getattr(model, name))
            dist.broadcast(t, src=0)
    item_f = sp.identity(n_items, dtype=np.float32, format="csr")
    user_f = sp.identity(n_users, dtype=np.float32, format="csr")
    positives = model._get_positives_lookup_matrix(train)
    fl = model._get_lightfm_data()
    session = _Session(fl, CSRMatrix(item_f), CSRMatrix(user_f), device=local_rank)
    lookup = positives
    if args.debug_empty_positives:
        lookup = sp.csr_matrix(positives.shape, dtype=np.float32)
    session.set_interactions(CSRMatrix(lookup), np.ascontiguousarray(train.row),
                             np.ascontiguousarray(train.col), train.data, train.data)
    total = args.warmup + args.steps
    seeds = []
    for e in range(total):  # lightfm.py:689-690 + _lightfm_fast.pyx.template:812-814
        shuffle = np.arange(train.nnz, dtype=np.int32)
        model.random_state.shuffle(shuffle)
        if args.debug_no_shuffle:
            shuffle = np.arange(train.nnz, dtype=np.int32)
        seeds.append(np.ascontiguousarray(model.random_state.randint(
            0, np.iinfo(np.int32).max, size=1).astype(np.uint32)))
        session.upload_shuffle(shuffle, slot=e)
    if world > 1:
        uid = C.create_string_buffer(N.UNIQUE_ID_BYTES)
        if rank == 0:
            N.check(N.lib().lfm_comm_unique_id(uid))
        import torch
        t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).clone()
        dist.broadcast(t, src=0)
        session.comm_init(C.create_string_buffer(bytes(t.numpy().tobytes()), N.UNIQUE_ID_BYTES), rank, world)
    log("setup done in %.1fs on %s (%d CUs)" % (time.time() - t0, dev_name, cus))

    from lightfm_amd._lightfm_fast import make_opts

    def step(e):
        opts, _ = make_opts()
        opts.history = e * train.nnz  # epoch e of one training run: epoch 0 ramps the concurrency up
        session.epoch("warp", 0.0, 0.0, 5, 10, seeds[e], opts, slot=e)
        if not session.check_finite():
            raise SystemExit("model diverged")
        return opts

    def barrier():
        if world > 1:
            session.comm_barrier()
            dist.barrier()

    for e in range(args.warmup):
        step(e)
    barrier()  # lfm_session_epoch / check_finite synchronise the session's stream before returning
    t_start = time.perf_counter()
    stats = [step(args.warmup + e) for e in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        cnt = torch.tensor([float(sum(s.counters[0] for s in stats))], dtype=torch.float64)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total_pos = float(cnt[0])
    else:
        total_pos = float(sum(s.counters[0] for s in stats))

    # roofline of the dominant kernel (the WARP epoch kernel), this rank
    kernel_s = sum(s.kernel_ms for s in stats) / 1e3
    lens = np.diff(positives.indptr)[train.row]
    mean_probe = float(np.mean(8 + 4 * np.ceil(np.log2(lens + 1.0))))
    alg = algorithmic_bytes(sum(s.counters[0] for s in stats), sum(s.counters[1] for s in stats),
                            sum(s.counters[2] for s in stats), sum(s.counters[3] for s in stats),
                            D, 1, 1, mean_probe)
    launches = sum(int(s.launches) for s in stats)
    ng = int(stats[-1].tile_ng)
    kernel_name = ("fit_warp_kernel<1, true, 1>" if ng == 0 else
                   "fit_warp_tile_kernel<%d, %d, false, false>" % (64 // ng, {4: 4, 2: 2, 1: 1}[ng]))
    achieved = alg / kernel_s / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if os.path.exists(pmc_path):
        try:
            traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": kernel_name, "algorithmic_bytes_per_launch": alg / launches,
                "avg_launch_ms": kernel_s * 1e3 / launches, "launches_per_epoch": launches / args.steps,
                "interactions_per_wavefront_pass": ng, "interactions_in_flight": int(stats[-1].in_flight),
                "draws_per_interaction": sum(s.counters[1] for s in stats) / max(1.0, sum(s.counters[0] for s in stats)),
                "updates_per_interaction": sum(s.counters[2] for s in stats) / max(1.0, sum(s.counters[0] for s in stats))}

    if options.warp_kernel == 2:  # profiling build: per-phase shader cycles per wavefront pass
        ph = np.sum([list(s.phase_cycles) for s in stats], axis=0).astype(np.float64)
        passes = sum(s.counters[0] for s in stats) / float(max(1, int(stats[-1].tile_ng)))
        roofline["phase_cycles_per_pass"] = dict(zip(
            ("head", "gather", "score", "lookup", "acc_loads", "update", "tail", "unused"),
            [round(float(x) / passes, 1) for x in ph]))
    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(train, log)

    session.close()
    if rank == 0:
        value = total_pos / elapsed
        out = {
            "metric": "positive interactions/sec/epoch (WARP, ML-20M)",
            "value": value, "unit": "interactions/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "MovieLens-20M shape (138493 users x 26744 items x %d interactions per "
                                   "GPU), loss=warp, no_components=64, identity features, adagrad, "
                                   "max_sampled=10, one epoch per step" % train.nnz,
                       "parallelism": "rows sharded over %d GPU(s); item tables all-reduced per epoch" % args.gpus,
                       "device": dev_name},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if tuned:
            out["config"]["non_default_options"] = tuned
        if cpu:
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
