"""Dry run of bench.py's N > 1 control flow on CPU (world size 2 over gloo): the row sharding, the
rendezvous of the RCCL unique id, the segment / merge schedule every rank must walk in lockstep,
the barriers, the max-over-ranks timing and the contract line -- with the device session replaced
by a stand-in that counts calls (no kernel runs here; the arithmetic of the merges is covered by
tests/test_multi_gpu_semantics.py and, on the GPU, tests/test_hip_round2.py).  A rank that made a
different number of collective calls than the other would hang this test."""
import json
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _FakeLib(object):
    def lfm_comm_unique_id(self, buf):
        buf.raw = bytes(range(128))
        return 0


def _worker(rank, world, port, out, argv):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import lightfm_amd._native as N
    import lightfm_amd.lightfm as L

    calls = {"epoch": 0, "merge": 0, "positions": 0, "ids": []}

    class FakeSession(object):
        def __init__(self, struct, item_f, user_f, device=0):
            self.n = 0
            self.users = struct.user_features.shape[0]

        def set_interactions(self, positives, rows, cols, data, weight):
            assert rows.dtype == np.int32 and cols.dtype == np.int32 and data.dtype == np.float32
            assert rows.max() < self.users, "user ids must be relative to the rank's row range"
            self.n = len(rows)

        def build_positives(self, n_users, n_items):
            assert n_users == self.users

        def comm_init(self, uid, rank_, nranks):
            calls["ids"].append(bytes(uid.raw))
            assert (rank_, nranks) == (rank, world)

        def device_shuffle(self, k0, k1, slot=0):
            pass

        def epoch(self, loss, ia, ua, k, n, seeds, opts, slot=0):
            b, e = int(opts.pos_begin), int(opts.pos_end)
            assert 0 <= b < e <= self.n
            calls["epoch"] += 1
            calls["positions"] += e - b
            opts.counters[0] = e - b
            opts.counters[1] = 7 * (e - b)
            opts.counters[2] = (e - b) // 2
            opts.counters[3] = (e - b) // 2
            opts.kernel_ms = 1e-3 * (e - b)
            opts.launches, opts.tile_ng, opts.kernel_used, opts.in_flight = 1, 4, 1, 12288

        def comm_merge(self, sides, mode):
            calls["merge"] += 1
            dist.barrier()  # a collective: all ranks must call it the same number of times

        def comm_merge_sparse(self, sides, mode, overlap=True):
            calls["merge"] += 1
            self.pending = bool(overlap)
            dist.barrier()
            return 1000

        def comm_merge_hot(self, sides, mode, overlap=True):
            return self.comm_merge_sparse(sides, mode, overlap)

        def set_hot_rows(self, side, rows):
            calls["hot"] = len(rows)

        def comm_merge_flush(self):
            self.pending = False

        def check_finite(self):
            assert not getattr(self, "pending", False), "an overlapped exchange must be flushed before the tables are read"
            return True

        def comm_any(self, flag):
            import torch
            t = torch.tensor([int(bool(flag))])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return bool(t[0])

        def comm_barrier(self):
            dist.barrier()

        def close(self):
            pass

    L._Session = FakeSession
    N.device_count = lambda: 8
    N.preload_comm = lambda: True
    N.device_info = lambda i: ("fake-gfx950", 256, 288 << 30)
    N.lib = lambda: _FakeLib()
    N.check = lambda rc: rc
    sys.argv = ["bench.py"] + argv
    import io
    import contextlib
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_dryrun", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    with open(out % rank, "w") as f:
        json.dump({"stdout": buf.getvalue(), "calls": {k: v for k, v in calls.items() if k != "ids"},
                   "same_id": len(set(calls["ids"])) == 1, "id0": calls["ids"][0][:8].hex()}, f)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("scaling,config", [("strong", "c2"), ("weak", "c2"), ("strong", "c3")])
def test_bench_two_ranks_walk_the_same_schedule(tmp_path, scaling, config):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    out = str(tmp_path / "rank%d.json")
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--epochs-per-step", "1", "--config", config,
            "--scale", "0.004", "--scaling", scaling]
    mp.spawn(_worker, args=(world, port, out, argv), nprocs=world, join=True)
    got = [json.load(open(out % r)) for r in range(world)]
    assert got[0]["calls"]["merge"] == got[1]["calls"]["merge"] > 0
    assert got[0]["id0"] == got[1]["id0"], "both ranks must hold rank 0's unique id"
    assert got[1]["stdout"].strip() == ""  # only rank 0 prints
    line = json.loads(got[0]["stdout"].strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == scaling and line["steps"] == 2
    assert line["unit"] == "interactions/s" and line["higher_is_better"] is True
    # value = positives visited by ALL ranks in the timed steps / max-over-ranks time
    timed = sum(g["calls"]["positions"] for g in got)  # warm-up + timed positions of both ranks
    assert 0 < line["value"] * line["ms_per_step"] * 1e-3 * line["steps"] <= timed
    assert "roofline" in line and line["roofline"]["frac"] > 0 and line["cpu_baseline"] is None
    assert "GPUs" in line["config"]["parallelism"] and "rows touched" in line["config"]["parallelism"]
    if config == "c3":  # the hybrid config runs through the product driver: its tag rows are hot rows on both ranks
        assert got[0]["calls"].get("hot", 0) == got[1]["calls"].get("hot", 0) > 1000
        assert "hot rows" in line["config"]["parallelism"]


def _fit_worker(rank, world, port, out, precut=False, hybrid=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import lightfm_amd._native as N
        import lightfm_amd.lightfm as L
        from lightfm_amd import LightFM, synthetic
        from lightfm_amd.distributed import DistributedFit
        log = {"epoch": 0, "merge": 0, "positions": 0, "sides": [], "hot": {}, "kinds": []}

        class FakeSession(object):
            def __init__(self, struct, item_f, user_f, device=0):
                self.users = user_f.rows
                self.struct = struct
                log["shapes"] = [item_f.rows, item_f.cols, user_f.rows, user_f.cols,
                                 int(struct.item_features.shape[0]), int(struct.user_features.shape[0])]

            def set_interactions(self, positives, rows, cols, data, weight):
                assert rows.dtype == np.int32 and data.dtype == np.float32 and rows.max() < self.users
                self.n = len(rows)

            def build_positives(self, n_users, n_items):
                pass

            def comm_init(self, uid, r, n):
                pass

            def device_shuffle(self, k0, k1, slot=0):
                pass

            def epoch(self, loss, ia, ua, k, n, seeds, opts, slot=0):
                log["epoch"] += 1
                log["positions"] += int(opts.pos_end - opts.pos_begin)
                self.struct.user_features[:] += 1.0  # "training": every own user row moves

            def comm_merge(self, sides, mode):
                log["merge"] += 1
                dist.barrier()

            def comm_merge_sparse(self, sides, mode, overlap=True):
                log["merge"] += 1
                log["sides"].append(sides)
                log["kinds"].append("full")
                self.pending = bool(overlap)
                dist.barrier()
                return 1000

            def comm_merge_hot(self, sides, mode, overlap=True):
                log["merge"] += 1
                log["sides"].append(sides)
                log["kinds"].append("hot")
                dist.barrier()
                return 100

            def set_hot_rows(self, side, rows):
                assert np.all(np.diff(rows) > 0)
                log["hot"][str(side)] = [int(rows[0]), int(rows[-1]), len(rows)]

            def comm_merge_flush(self):
                self.pending = False

            def check_finite(self):
                assert not getattr(self, "pending", False), "flush before the finite check"
                return True

            def comm_any(self, flag):
                return bool(flag)

            def sync_to_host(self, struct):
                assert not getattr(self, "pending", False), "flush before the download"

            def close(self):
                pass

        L._Session = FakeSession
        N.lib = lambda: _FakeLib()
        N.check = lambda rc: rc
        data = synthetic.make_interactions(400, 300, 20000, seed=3).astype(np.float64)  # wrong dtype on purpose
        model = LightFM(no_components=8, loss="warp", random_state=5)
        if hybrid:
            # a hybrid model through the product driver: [identity | 20 tags] item features (the tags are hot
            # rows) and 12 shared user-feature columns (the user tables are then replicated and merged too)
            from tests import helpers as H
            item_f = H.tag_features(300, 20, 3, seed=1)
            user_f = H.tag_features(400, 12, 2, seed=2, with_identity=False)
            from lightfm_amd.distributed import MergePolicy
            fit = DistributedFit(model, data, rank, world, device=rank, dist=dist, item_features=item_f,
                                 user_features=user_f, policy=MergePolicy(hot_max=2048))  # (default: every world * 2**17)
            assert model.item_embeddings.shape[0] == 320 and model.user_embeddings.shape[0] == 12
        elif precut:
            # a rank that only holds its own users' rows (its range of a large job's data source)
            from lightfm_amd.distributed import local_shard, plan_row_shards
            coo = data.tocoo()
            bounds = plan_row_shards(np.bincount(coo.row, minlength=coo.shape[0]), world)
            mine, _ = local_shard(coo, rank, world, bounds=bounds)
            assert 0 < mine.nnz < coo.nnz
            fit = DistributedFit(model, mine, rank, world, device=rank, dist=dist, bounds=bounds, global_n=coo.nnz)
        else:
            fit = DistributedFit(model, data, rank, world, device=rank, dist=dist)
        before = model.user_embeddings.copy()
        fit.run(epochs=2)
        fit.gather_users()
        fit.close()
        b0, b1 = fit.user_range
        json.dump({"log": log, "n_local": int(fit.shard.nnz), "range": [b0, b1], "global_n": int(fit.global_n),
                   "sides": int(fit.sides), "moved_all": bool(np.all(model.user_embeddings != before))}, open(out % rank, "w"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("precut", [False, True])
def test_distributed_fit_control_flow_two_ranks(tmp_path, precut):
    """DistributedFit (the class a multi-GPU user drives) with a stand-in session: dtype coercion of the
    interactions, disjoint contiguous user ranges covering every user, every local position trained once per
    epoch, the same number of merges on both ranks, and gather_users() delivering the other rank's rows."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    out = str(tmp_path / "fit%d.json")
    mp.spawn(_fit_worker, args=(world, port, out, precut), nprocs=world, join=True)
    got = [json.load(open(out % r)) for r in range(world)]
    assert got[0]["log"]["merge"] == got[1]["log"]["merge"] > 2
    assert got[0]["range"][0] == 0 and got[0]["range"][1] == got[1]["range"][0] and got[1]["range"][1] == 400
    for g in got:
        assert g["log"]["positions"] == 2 * g["n_local"]
        assert g["moved_all"], "after gather_users every rank holds the trained rows of every user"
    assert got[0]["n_local"] + got[1]["n_local"] == got[0]["global_n"] == got[1]["global_n"] > 15000


@pytest.mark.timeout(600)
def test_distributed_fit_hybrid_model_two_ranks(tmp_path):
    """BASELINE configs C3 / C5 are hybrid: DistributedFit takes item_features / user_features like the
    reference's fit_partial (LFM:560-666).  Item tables are replicated with the tag columns as hot rows;
    shared user features make the user tables replicated too (sides = 3) and every rank's session sees its
    users' rows of the matrix over ALL feature columns."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    out = str(tmp_path / "hyb%d.json")
    mp.spawn(_fit_worker, args=(world, port, out, False, True), nprocs=world, join=True)
    got = [json.load(open(out % r)) for r in range(world)]
    for g in got:
        assert g["sides"] == 3 and set(g["log"]["sides"]) == {3}
        assert g["log"]["hot"]["0"] == [300, 319, 20], g["log"]["hot"]       # the 20 tag columns behind the identity block
        assert g["log"]["hot"]["1"][2] == 12                                  # every shared user-feature column
        assert "hot" in g["log"]["kinds"] and "full" in g["log"]["kinds"]
        n_local_users = g["range"][1] - g["range"][0]
        assert g["log"]["shapes"] == [300, 320, n_local_users, 12, 320, 12]
        assert g["log"]["positions"] == 2 * g["n_local"]
    assert got[0]["log"]["kinds"] == got[1]["log"]["kinds"]                   # the same schedule on both ranks


def _owner_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import lightfm_amd.lightfm as L
        from lightfm_amd import LightFM, synthetic
        from lightfm_amd.distributed import DistributedFit
        log = {"calls": [], "epochs": 0}

        class FakeSession(object):
            handle = True

            def __init__(self, struct, item_f, user_f, device=0):
                self.struct = struct

            def set_interactions(self, positives, rows, cols, data, weight):
                self.n = len(rows)

            def build_positives(self, n_users, n_items):
                pass

            def export_items(self):
                log["calls"].append("export")
                return bytes([rank]) * 336

            def share_items_ipc(self, exports, my_rank):
                assert my_rank == rank and len(exports) == world
                assert [e[0] for e in exports] == list(range(world)), "every rank's export, in rank order"
                log["calls"].append("share")

            def device_shuffle(self, k0, k1, slot=0):
                pass

            def epoch(self, loss, ia, ua, k, n, seeds, opts, slot=0):
                assert int(opts.pos_begin) == 0 and int(opts.pos_end) == self.n, "owner-sharded: no segments, no merges"
                log["epochs"] += 1
                self.struct.item_features[rank::world] += 1.0   # "training" of the rows this rank owns

            def check_finite(self):
                return True

            def gather_shared_items(self):
                log["calls"].append("gather")

            def sync_to_host(self, struct):
                log["calls"].append("sync")

            def comm_merge_sparse(self, *a, **k):
                raise AssertionError("owner-sharded item tables are never merged")

            comm_init = comm_merge = comm_merge_hot = comm_merge_sparse

            def close(self):
                log["calls"].append("close")
                self.handle = None

        L._Session = FakeSession
        data = synthetic.make_interactions(300, 200, 12000, seed=3)
        model = LightFM(no_components=16, loss="warp", random_state=5)
        fit = DistributedFit(model, data, rank, world, device=0, dist=dist, item_tables="owner")
        assert fit.owner_sharded
        fit.run(epochs=2)
        fit.close()
        # refused outside the kernel's scope (every rank raises before any collective)
        for bad in (dict(loss="bpr"), dict(loss="warp", no_components=96), dict(loss="warp", item_alpha=1e-6)):
            kw = dict(no_components=16, loss="warp", random_state=5)
            kw.update(bad)
            with pytest.raises(NotImplementedError):
                DistributedFit(LightFM(**kw), data, rank, world, device=0, dist=dist, item_tables="owner")
        json.dump({"log": log, "merges": fit.merges}, open(out % rank, "w"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_distributed_fit_owner_sharded_control_flow_two_ranks(tmp_path):
    """DistributedFit(item_tables="owner") with a stand-in session over gloo: the exports of all ranks reach every rank in rank
    order, an epoch is ONE call over all local positions (no segments, no merges, no communicator), the finite flag and the final
    gather are collective (a rank that skipped one would hang this test), sessions close behind a barrier."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    out = str(tmp_path / "owner%d.json")
    mp.spawn(_owner_worker, args=(world, port, out), nprocs=world, join=True)
    got = [json.load(open(out % r)) for r in range(world)]
    for g in got:
        assert g["merges"] == 0 and g["log"]["epochs"] == 2
        assert g["log"]["calls"] == ["export", "share", "gather", "sync", "close"], g["log"]["calls"]
