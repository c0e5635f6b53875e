"""Multi-GPU host logic (no counterpart in the reference).

One process per GPU.  The interaction matrix is sharded by USER into contiguous row ranges
balanced by interaction count.  With identity user features a rank holds ONLY its own users'
rows of the user tables (ids rebased to the range; never communicated); the item tables are
replicated.  An epoch runs as SEGMENTS of the rank's shuffled shard with a merge of the
replicated tables after every segment (csrc/session.hip: merge_group_sparse -- the rows that changed
since the last merge, detected at merge time and OR-ed over the ranks, travel as packed deltas through
an RCCL all-reduce over xGMI on a communication stream of their own.  The merge is SYNCHRONOUS by
default; `MergePolicy.overlap` lets the exchange of segment j run under the kernels of segment j + 1
and land at the next merge, which was measured to cost precision@10 and is off):

* every rank derives the same segment list from global numbers only (`merge_schedule`), so all
  ranks call the collective the same number of times;
* the interval between merges GROWS with the training history -- like the number of
  interactions in flight inside one GPU (DESIGN.md "Hogwild at GPU width"), replicas that
  have not exchanged their updates are harmless once the model has left its initial state and
  ruinous before -- from `merge_min` interactions up to `merge_max` (all ranks together);
* HOT rows -- feature rows shared by many items (the tag rows of a hybrid model) -- are merged at
  a short cadence of their own between the full merges (`MergePolicy.hot_share / hot_max`,
  `hot_rows`): they are few (a megabyte) and every interaction of every rank updates some.

`torch.distributed` (any backend, gloo is enough) is used ONLY for the rendezvous: broadcasting
the RCCL unique id and the initial tables, and gathering the user rows at the end.  The data
path never goes through it.
"""
import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

from . import _native as N

__all__ = ["plan_row_shards", "local_shard", "rank_seed", "merge_deltas", "merge_schedule", "merge_plan",
           "hot_rows", "MergePolicy", "DistributedFit"]


def plan_row_shards(row_counts, world):
    """Boundaries b[0..world] of contiguous user ranges [b[r], b[r+1]) whose interaction counts
    are as equal as contiguous ranges allow (greedy split of the prefix sum)."""
    row_counts = np.asarray(row_counts, dtype=np.int64)
    n_rows = len(row_counts)
    csum = np.concatenate([[0], np.cumsum(row_counts)])
    total = csum[-1]
    bounds = [0]
    for r in range(1, world):
        target = total * r // world
        b = int(np.searchsorted(csum, target, side="left"))
        bounds.append(min(max(b, bounds[-1]), n_rows))
    bounds.append(n_rows)
    return np.asarray(bounds, dtype=np.int64)


def local_shard(interactions, rank, world, bounds=None, rebase=False):
    """The rank's interactions as a COO.  rebase=False: full shape, global user ids.
    rebase=True: shape (users of the rank, n_items), user ids relative to bounds[rank]."""
    coo = interactions.tocoo()
    if bounds is None:
        bounds = plan_row_shards(np.bincount(coo.row, minlength=coo.shape[0]), world)
    keep = (coo.row >= bounds[rank]) & (coo.row < bounds[rank + 1])
    rows, shape = coo.row[keep], coo.shape
    if rebase:
        rows = (rows - bounds[rank]).astype(coo.row.dtype)
        shape = (int(bounds[rank + 1] - bounds[rank]), coo.shape[1])
    return sp.coo_matrix((coo.data[keep], (rows, coo.col[keep])), shape=shape,
                         dtype=coo.dtype), bounds


def rank_seed(seed, rank):
    """Per-rank RandomState seed: every rank shuffles its own shard and draws its own kernel
    seeds, while rank 0 keeps the caller's stream (a 1-GPU run is unchanged)."""
    return int((int(seed) + 0x9E3779B1 * int(rank)) % (2 ** 32))


def merge_deltas(start, local, all_reduce_sum):
    """X := X_start + sum over ranks (X_rank - X_start): the LFM_MERGE_SUM arithmetic of
    csrc/session.hip restated in numpy (tests; `all_reduce_sum` sums an array over ranks)."""
    delta = (local - start).astype(np.float32)
    return (start + all_reduce_sum(delta)).astype(np.float32)


class MergePolicy(object):
    """When and how the replicated tables are merged.

    merge_k    the interval between merges is (interactions all ranks have trained on so far)
               / merge_k, clamped to [lo, cap]
    merge_min  lo = max(merge_min, rows_lo * replicated rows) interactions of ALL ranks together
    merge_max  cap; 0 = max(world * 2**20 (one full-size launch per rank), rows_k * replicated rows)
    rows_lo, rows_k  optional: scale the bounds with the replicated table (interactions per ROW the
               shortest / longest interval may span).  0 (default) = off: a 1/8-scale C4 emulated at 8
               ranks lost 0.004 precision@10 with rows_k = 64 (two merges per epoch) against 0.002 at
               the plain world * 2**20 cap -- a trade of quality for exchange volume the caller makes
    mode       "sum" | "mean" | "adagrad" (include/lfm_hip.h: LFM_MERGE_*)
    sparse     exchange only the rows touched since the last merge (default); False = the dense
               all-reduce of whole tables
    overlap    sparse merges: the exchange overlaps the next segment and lands one merge later.
               Off by default: measured with 8 emulated ranks it costs precision@10 (C2 -0.007 at the
               8 Mi cap, -0.002 at half of it), and what it hides is small next to the kernels wherever
               the quality-preserving cadence is affordable at all (DESIGN.md "Multi-GPU")
    hot_share  a feature column of the replicated side that an interaction touches with at least this
               probability -- 2 nnz(column) / rows of the feature matrix (the positive and the negative
               item), columns with a single entry (identity blocks) never -- is a HOT row: the tag / genre
               rows shared by many items (C3: 0.014; ML-20M identity rows: 0.00007; C5's hashed rows:
               0.000016).  Hot rows are merged between the full merges at the short cadence hot_max
               (0 = world * 2**17 interactions)
    """

    def __init__(self, merge_k=4, merge_min=16384, merge_max=0, mode="adagrad", rows_lo=0, rows_k=0,
                 sparse=True, overlap=False, hot_share=1.0 / 512, hot_max=0):
        self.merge_k, self.merge_min, self.merge_max, self.mode = merge_k, merge_min, merge_max, mode
        self.rows_lo, self.rows_k, self.sparse, self.overlap = rows_lo, rows_k, sparse, overlap
        self.hot_share, self.hot_max = hot_share, hot_max

    def mode_id(self):
        return N.MERGE_MODES[self.mode]


def merge_schedule(global_history, global_n, world, policy=None, n_rows=0):
    """Segment boundaries of one epoch as FRACTIONS of the epoch, 0 = f[0] < ... < f[-1] = 1,
    computed from global numbers only (identical on every rank).  A rank runs its shuffled
    positions [round(f[j] * n_local), round(f[j+1] * n_local)) and merges after each.
    n_rows = rows of the replicated tables (0: the bounds do not scale with the table)."""
    policy = policy or MergePolicy()
    if global_n <= 0:
        return np.array([0.0, 1.0])
    cap = policy.merge_max if policy.merge_max > 0 else max(world * (1 << 20), policy.rows_k * int(n_rows))
    lo = max(1, min(max(policy.merge_min, policy.rows_lo * int(n_rows)), cap))
    fr, g = [0.0], 0
    while g < global_n:
        seg = int(min(cap, max(lo, (global_history + g) // max(1, policy.merge_k))))
        g = min(global_n, g + seg)
        fr.append(g / float(global_n))
    fr[-1] = 1.0
    return np.asarray(fr)


def hot_rows(features, hot_share, touches=2.0):
    """Ascending feature columns of a (replicated side's) feature matrix that an interaction touches with
    probability >= hot_share (MergePolicy.hot_share); none for an identity matrix.  touches = rows of the
    matrix an interaction reads: 2 on the item side (the positive and the negative item), 1 on the user side."""
    if features is None or hot_share <= 0 or features.shape[0] == 0:
        return np.zeros(0, np.int32)
    f = sp.csc_matrix(features)
    counts = np.diff(f.indptr)
    return np.flatnonzero((counts >= 2) & (touches * counts / float(features.shape[0]) >= hot_share)).astype(np.int32)


def merge_plan(global_history, global_n, world, policy=None, n_rows=0, has_hot=False):
    """(fractions, kinds): the full-merge schedule of `merge_schedule`, its segments cut further into pieces
    of at most policy.hot_max interactions when there are hot rows; kinds[j] = "full" | "hot" names the merge
    after segment j.  Identical on every rank."""
    policy = policy or MergePolicy()
    full = merge_schedule(global_history, global_n, world, policy, n_rows)
    if not has_hot or global_n <= 0:
        return full, ["full"] * (len(full) - 1)
    hot_cap = policy.hot_max if policy.hot_max > 0 else world * (1 << 17)
    fr, kinds = [0.0], []
    for j in range(len(full) - 1):
        a, b = full[j], full[j + 1]
        pieces = max(1, int(np.ceil((b - a) * global_n / float(hot_cap))))
        for q in range(1, pieces + 1):
            fr.append(a + (b - a) * q / pieces)
            kinds.append("hot" if q < pieces else "full")
    fr[-1] = 1.0
    return np.asarray(fr), kinds


def segment_positions(fractions, n_local):
    """The rank's positions for the schedule's fractions (monotone, ends at n_local)."""
    pos = np.rint(np.asarray(fractions) * n_local).astype(np.int64)
    pos[0], pos[-1] = 0, n_local
    return np.maximum.accumulate(pos)


class DistributedFit(object):
    """Drives the epochs of one rank.  Usage (inside torch.distributed.run):

        fit = DistributedFit(model, interactions, rank, world, device=local_rank, dist=dist,
                             item_features=item_features, user_features=user_features)
        fit.run(epochs)
        fit.gather_users()       # optional: every rank's model then holds all user rows

    `model` is a lightfm_amd.LightFM; `item_features` / `user_features` are what the reference's
    `fit_partial` takes (LFM:560-666; None = identity).  What is replicated and what is partitioned:

    * item side: the tables have one row per item FEATURE and every rank's negatives range over all
      items (PYX:860-861) -- always replicated, merged over RCCL; feature columns shared by many items (a
      hybrid model's tag rows) are HOT rows with a short merge cadence of their own (`hot_rows`);
    * user side, identity features: a user's row is touched only by that user's interactions, i.e. by its
      owner rank -- the tables are PARTITIONED (a rank allocates only its range; never communicated; the
      host arrays keep the full shape and `gather_users` fills them in after training);
    * user side with a feature matrix: feature rows are shared between users of different ranks --
      replicated and merged like the item side (`sides` bit 1), the rank's session sees the rows of the
      matrix that belong to its users.
    """

    def __init__(self, model, interactions, rank, world, device=0, dist=None, policy=None,
                 host_shuffle=False, bounds=None, global_n=None, item_features=None, user_features=None,
                 local_ids=False, item_tables="replicated"):
        """item_tables: "replicated" (default: every rank holds the item tables, merged over RCCL) or "owner"
        (owner-sharded, include/lfm_hip.h: lfm_session_share_items_ipc -- the item rows are cut into `world`
        contiguous ranges, range r lives with rank r, and every rank's kernels gather from and publish to the
        owner's memory through HIP IPC mappings: no replicas, no merges, no RCCL; for item sides too large to
        merge at a useful cadence, BASELINE config C4.  Parallel WARP, identity features on both sides, adagrad,
        no regularisation, no_components <= 64 and a multiple of 4, max_sampled = 10).

        interactions: the WHOLE interaction matrix (every rank cuts its own user range from it,
        boundaries planned from the row counts) -- or, with `bounds` and `global_n`, only THIS rank's
        rows of it: a matrix of the full shape whose entries all lie in users [bounds[rank],
        bounds[rank + 1]) (a rank of a large job loads its range from its own data source; bounds =
        the user boundaries of all ranks, global_n = interactions of all ranks together) -- or, with
        `local_ids=True` and `global_n`, this rank's shard with user ids already relative to its range
        (shape (users of the rank, n_items); user_features, if any, are the rank's rows; the model then
        holds ONLY the rank's user rows on the host as well: what a 50 M-user job needs, where no process
        can hold the full user tables; `gather_users` is not available)."""
        from ._lightfm_fast import CSRMatrix, FastLightFM
        from .lightfm import _Session, _WEIGHTS
        self.model, self.rank, self.world, self.dist = model, rank, world, dist
        self.policy = policy or MergePolicy()
        self.host_shuffle = host_shuffle
        if item_tables not in ("replicated", "owner"):
            raise ValueError("item_tables must be 'replicated' or 'owner'")
        self.owner_sharded = item_tables == "owner" and world > 1
        if self.owner_sharded and (item_features is not None or user_features is not None or model.loss != "warp"
                                   or model.learning_schedule != "adagrad" or model.item_alpha or model.user_alpha
                                   or model.no_components > 64 or model.no_components % 4 or model.max_sampled != 10):
            raise NotImplementedError("owner-sharded item tables: parallel WARP with identity features, adagrad, no "
                                      "regularisation, no_components <= 64 (a multiple of 4), max_sampled = 10")
        coo = interactions.tocoo()
        coo = sp.coo_matrix((np.ascontiguousarray(coo.data, dtype=np.float32),
                             (np.ascontiguousarray(coo.row, dtype=np.int32),
                              np.ascontiguousarray(coo.col, dtype=np.int32))), shape=coo.shape)
        self.local_ids = bool(local_ids)
        if local_ids:
            if global_n is None or bounds is not None:
                raise ValueError("local_ids needs global_n (interactions of all ranks together) and no bounds")
            shard = coo
            self.bounds = None
            self.global_n = int(global_n)
        elif bounds is not None:
            if global_n is None:
                raise ValueError("a pre-cut shard needs global_n (interactions of all ranks together)")
            bounds = np.asarray(bounds, dtype=np.int64)
            if len(bounds) != world + 1 or bounds[0] != 0 or bounds[-1] != coo.shape[0] or np.any(np.diff(bounds) < 0):
                raise ValueError("bounds must be the %d user boundaries 0 = b[0] <= ... <= b[world] = n_users" % (world + 1))
            if coo.nnz and (coo.row.min() < bounds[rank] or coo.row.max() >= bounds[rank + 1]):
                raise ValueError("a pre-cut shard may only hold users of this rank's range")
            shard, self.bounds = local_shard(coo, rank, world, bounds=bounds, rebase=True)
            self.global_n = int(global_n)
        else:
            shard, self.bounds = local_shard(coo, rank, world, rebase=True)
            self.global_n = int(coo.nnz)
        self.shard = shard
        n_users, n_items = coo.shape
        # the reference's checks and coercions of the feature matrices (LFM:314-363)
        user_f, item_f = model._construct_feature_matrices(n_users, n_items, user_features, item_features)
        self.shared_users = user_features is not None
        if model.item_embeddings is None:
            model._initialize(model.no_components, item_f.shape[1], user_f.shape[1])
        if not item_f.shape[1] == model.item_embeddings.shape[0]:
            raise ValueError("Incorrect number of features in item_features")
        if not user_f.shape[1] == model.user_embeddings.shape[0]:
            raise ValueError("Incorrect number of features in user_features")
        if world > 1:  # replicas start from rank 0's tables (with local ids the user rows are each rank's own)
            import torch
            for name in ("item_embeddings",) + (() if (local_ids and not self.shared_users) else ("user_embeddings",)):
                dist.broadcast(torch.from_numpy(getattr(model, name)), src=0)
        b0, b1 = (0, n_users) if local_ids else (int(self.bounds[rank]), int(self.bounds[rank + 1]))
        self.user_range = (b0, b1)
        arrays = []
        for name in _WEIGHTS:  # views: device results land in the model's own arrays
            a = getattr(model, name)
            arrays.append(a[b0:b1] if (name.startswith("user") and not self.shared_users) else a)
        self.struct = FastLightFM(*arrays, model.no_components,
                                  int(model.learning_schedule == "adadelta"), model.learning_rate,
                                  model.rho, model.epsilon, model.max_sampled)
        if self.shared_users:
            # the rank's users' rows of the matrix over ALL user-feature columns (replicated tables)
            rank_user_f = user_f[b0:b1].tocsr()
            rank_user_f.sort_indices()
        else:
            rank_user_f = sp.identity(b1 - b0, dtype=np.float32, format="csr")
        self.item_features, self.user_features = item_f, rank_user_f
        # The ranks exchange item rows: they must agree on the device row stride, which a one-GPU session pads by what ITS
        # feature matrices look like (csrc/session.hip, LIGHTFM_AMD_ROW_ALIGN) -- a rank's slice of the user features can
        # look like an identity on one rank only.  So a rank's session takes the width-only rule (mode 2) unless padding is off.
        align = os.environ.get("LIGHTFM_AMD_ROW_ALIGN")
        if align != "0":
            os.environ["LIGHTFM_AMD_ROW_ALIGN"] = "2"
        try:
            self.session = _Session(self.struct, CSRMatrix(item_f), CSRMatrix(rank_user_f), device=device)
        finally:
            if align is None:
                os.environ.pop("LIGHTFM_AMD_ROW_ALIGN", None)
            else:
                os.environ["LIGHTFM_AMD_ROW_ALIGN"] = align
        # LFM:381-386: without a sample_weight matrix every interaction weighs 1 -- the VALUES are Y (logistic's labels; a
        # positive wherever > 0), not weights
        weights = shard.data if model._scan(shard.data)[0] else np.ones_like(shard.data)
        self.session.set_interactions(None, np.ascontiguousarray(shard.row),
                                      np.ascontiguousarray(shard.col), shard.data, weights)
        self.session.build_positives(b1 - b0, n_items)
        # what the merges cover: bit 0 = the item tables, bit 1 = user tables of shared user features
        self.sides = 1 | (2 if self.shared_users else 0)
        self.n_replicated_rows = max(item_f.shape[1], user_f.shape[1] if self.shared_users else 0)
        self.merges, self.merge_bytes = 0, 0
        self.hot = [hot_rows(item_features, self.policy.hot_share, 2.0),
                    hot_rows(user_features if self.shared_users else None, self.policy.hot_share, 1.0)]
        if self.owner_sharded:
            # every rank's export to every rank (the rendezvous plane: CPU tensors, i.e. a gloo-capable process group),
            # then the peers' allocations are mapped.  A rank-local failure (hipIpcOpenMemHandle, out of memory) must not
            # leave the others blocked in the next rendezvous: every step ends with an agreement on an error flag
            # (_agree), and all ranks raise together.
            import torch
            err = None
            try:
                mine = torch.frombuffer(bytearray(self.session.export_items()), dtype=torch.uint8).clone()
            except Exception as exc:  # noqa: BLE001 -- reported on every rank below
                err, mine = exc, torch.zeros(C.sizeof(N.LfmItemExport), dtype=torch.uint8)
            self._agree(err, "exporting the item tables")
            blobs = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(blobs, mine)
            try:
                self.session.share_items_ipc([bytes(b.numpy().tobytes()) for b in blobs], rank)
            except Exception as exc:  # noqa: BLE001
                err = exc
            self._agree(err, "mapping the owners' item tables")  # nobody trains before every rank has its mappings
        elif world > 1:
            import torch
            uid = C.create_string_buffer(N.UNIQUE_ID_BYTES)
            if rank == 0:
                N.check(N.lib().lfm_comm_unique_id(uid))
            t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).clone()
            dist.broadcast(t, src=0)
            self.session.comm_init(C.create_string_buffer(bytes(t.numpy().tobytes()),
                                                          N.UNIQUE_ID_BYTES), rank, world)
            for side in (0, 1):
                if len(self.hot[side]):
                    self.session.set_hot_rows(side, self.hot[side])

    @property
    def has_hot(self):
        return len(self.hot[0]) > 0 or len(self.hot[1]) > 0

    def run_epoch(self, seeds, slot=0):
        """One epoch of this rank: segments + merges.  Returns the per-segment lfm_opts."""
        from ._lightfm_fast import make_opts
        m = self.model
        n = self.shard.nnz
        history = int(getattr(m, "_trained_interactions", 0))  # interactions of ALL ranks so far
        if self.world == 1 or self.owner_sharded:
            # nothing to merge: the whole epoch in one call, exactly LightFM.fit_partial's epoch (owner-sharded item
            # tables: all ranks train against ONE copy of every row -- plain Hogwild across the ranks)
            fr, kinds = np.array([0.0, 1.0]), ["none"]
        else:
            fr, kinds = merge_plan(history, self.global_n, self.world, self.policy, self.n_replicated_rows,
                                   self.has_hot)
        pos = segment_positions(fr, n)
        sparse = self.policy.sparse
        stats = []
        for j in range(len(pos) - 1):
            opts, _ = make_opts()
            # in-flight ramp: this rank's share of what all ranks have trained on
            opts.history = (history + int(round(self.global_n * pos[j] / max(1, n)))) // self.world
            opts.pos_begin, opts.pos_end = int(pos[j]), int(pos[j + 1])
            if pos[j + 1] > pos[j]:
                self.session.epoch(m.loss, m.item_alpha, m.user_alpha, m.k, m.n, seeds, opts, slot=slot)
            if kinds[j] == "none":
                pass
            elif not sparse:
                if kinds[j] == "full":
                    self.session.comm_merge(self.sides, self.policy.mode_id())
            elif kinds[j] == "hot":
                self.merge_bytes += self.session.comm_merge_hot(self.sides, self.policy.mode_id(), self.policy.overlap)
            else:
                self.merge_bytes += self.session.comm_merge_sparse(self.sides, self.policy.mode_id(), self.policy.overlap)
            self.merges += int(kinds[j] != "none")
            if pos[j + 1] > pos[j]:
                stats.append(opts)
        if sparse and self.world > 1 and not self.owner_sharded:
            self.session.comm_merge_flush()  # the last exchange lands before anything reads the tables
        m._trained_interactions = history + self.global_n
        return stats

    def epoch(self, num_threads=1):
        """One epoch as LightFM.fit_partial runs it (LFM:654-664): the shuffle, the kernel seeds, the
        segments with their merges, the finite check -- all ranks raise together (a rank that stopped
        alone would leave the others blocked in the next collective).  Returns the per-segment lfm_opts."""
        m = self.model
        n = self.shard.nnz
        from .options import options
        slot = getattr(self, "_slot", 0)
        if self.host_shuffle:
            shuffle = np.arange(n, dtype=np.int32)
            m.random_state.shuffle(shuffle)
            self.session.upload_shuffle(shuffle)
        elif getattr(self, "_ahead_keys", None) is None:
            keys = m.random_state.randint(0, np.iinfo(np.int32).max, size=624)
            self.session.device_shuffle(int(keys[0]), int(keys[1]), slot=slot)
        seeds = None
        if m.loss != "logistic":
            seeds = np.ascontiguousarray(m.random_state.randint(
                0, np.iinfo(np.int32).max, size=num_threads).astype(np.uint32))
        self._ahead_keys = None
        if not self.host_shuffle and options.shuffle_ahead and hasattr(self.session, "device_shuffle_ahead"):
            # the next epoch's permutation is written to the other slot while this epoch trains (the same keys -> seeds
            # -> keys order of draws; one block more is drawn after the last epoch of a run)
            self._ahead_keys = m.random_state.randint(0, np.iinfo(np.int32).max, size=624)
            self.session.device_shuffle_ahead(int(self._ahead_keys[0]), int(self._ahead_keys[1]), slot=1 - slot)
        stats = self.run_epoch(seeds, slot=slot)
        if self._ahead_keys is not None:
            self._slot = 1 - slot
        bad = not self.session.check_finite()  # (owner-sharded: the item rows this rank owns + its users)
        if self.owner_sharded:
            bad = self._any(bad)
        elif self.world > 1:
            bad = self.session.comm_any(bad)
        if bad:
            if self.owner_sharded:
                self._collect_items()
            self.session.sync_to_host(self.struct)
            raise ValueError("Not all estimated parameters are finite, your model may have diverged.")
        return stats

    def run(self, epochs, num_threads=1):
        stats = []
        for _ in range(epochs):
            stats.extend(self.epoch(num_threads))
        if self.owner_sharded:
            self._collect_items()
        self.session.sync_to_host(self.struct)
        return stats

    def _agree(self, error, what):
        """A rendezvous that doubles as the barrier of an owner-sharded step: every rank contributes whether ITS part
        failed; if any did, every rank raises (the failing ones their own exception) instead of some waiting forever
        in the next collective."""
        import torch
        t = torch.tensor([0 if error is None else 1], dtype=torch.int32)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        if error is not None:
            raise error
        if int(t[0]):
            raise RuntimeError("another rank failed while %s" % what)

    def _any(self, flag):
        """max over ranks of a flag on the rendezvous plane (owner-sharded jobs have no RCCL communicator)."""
        import torch
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return bool(int(t[0]))

    def _collect_items(self):
        """Owner-sharded item tables, after training: every rank copies the other owners' rows into its own
        tables (device to device through the mappings) -- between two barriers, so nobody trains meanwhile."""
        self.dist.barrier()
        err = None
        try:
            self.session.gather_shared_items()
        except Exception as exc:  # noqa: BLE001 -- all ranks raise together (_agree)
            err = exc
        self._agree(err, "gathering the item rows from their owners")

    def barrier(self):
        """Device work of this rank done and every rank here (bench.py's timed region)."""
        if self.owner_sharded:
            self.dist.barrier()  # (lfm_session_epoch returns with the rank's stream drained)
        elif self.world > 1:
            self.session.comm_barrier()

    def gather_users(self):
        """Every rank receives the other ranks' user rows (host plane, once, after training)."""
        if self.world <= 1 or self.shared_users:  # shared user features: the tables are replicated anyway
            return
        if self.local_ids:
            raise ValueError("gather_users needs the global user ranges (not available with local_ids)")
        import torch
        from .lightfm import _WEIGHTS
        for name in _WEIGHTS:
            if not name.startswith("user"):
                continue
            a = getattr(self.model, name)
            for r in range(self.world):
                b0, b1 = int(self.bounds[r]), int(self.bounds[r + 1])
                if b1 > b0:
                    self.dist.broadcast(torch.from_numpy(a[b0:b1]), src=r)

    def close(self):
        if self.owner_sharded and self.session.handle:
            # an owner's tables stay mapped in the peers until every rank is done with them; a rank that arrives here on an
            # error path of its own still takes part (monitored: a rank that died is reported after the timeout instead of
            # blocking the others for good)
            import datetime
            if hasattr(self.dist, "monitored_barrier"):  # (gloo: the owner-sharded rendezvous needs a gloo-capable group anyway)
                self.dist.monitored_barrier(timeout=datetime.timedelta(seconds=300))
            else:
                self.dist.barrier()
        self.session.close()
