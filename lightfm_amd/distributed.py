"""Multi-GPU host logic (no counterpart in the reference).

One process per GPU.  The interaction matrix is sharded by USER into contiguous row ranges
balanced by interaction count; every rank trains its shard on its own GPU with the replicated
item tables and its own slice of the user tables; after every epoch the item tables are merged
on device by an RCCL all-reduce of their deltas (csrc/session.hip: merge_side), and once at the
end the user tables are merged the same way (rows are disjoint across ranks with identity user
features, so that merge is an exact union).

`torch.distributed` (any backend, gloo is enough) is used ONLY for the rendezvous: broadcasting
the RCCL unique id and the initial item table.  The data path never goes through it.
"""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _native as N

__all__ = ["plan_row_shards", "local_shard", "rank_seed", "merge_deltas", "DistributedFit"]


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


def local_shard(interactions, rank, world, bounds=None):
    """The rank's interactions as a COO of the FULL shape (ids stay global)."""
    coo = interactions.tocoo()
    if bounds is None:
        bounds = plan_row_shards(np.bincount(coo.row, minlength=coo.shape[0]), world)
    keep = (coo.row >= bounds[rank]) & (coo.row < bounds[rank + 1])
    return sp.coo_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=coo.shape,
                         dtype=coo.dtype), bounds


def rank_seed(seed, rank):
    """Per-rank RandomState seed: every rank shuffles its own shard and draws its own kernel
    seeds, while rank 0 keeps the caller's stream (a 1-GPU run is unchanged)."""
    return int((int(seed) + 0x9E3779B1 * int(rank)) % (2 ** 32))


def merge_deltas(start, local, all_reduce_sum):
    """X := X_start + sum over ranks (X_rank - X_start): the merge csrc/session.hip performs on
    device with RCCL, restated in numpy (tests; `all_reduce_sum` sums an array over ranks)."""
    delta = (local - start).astype(np.float32)
    return (start + all_reduce_sum(delta)).astype(np.float32)


class DistributedFit(object):
    """Drives `LightFM`-compatible epochs of one rank.  Usage (inside torch.distributed.run):

        fit = DistributedFit(model, interactions, rank, world, device=local_rank, dist=dist)
        fit.run(epochs)          # model's weights are complete on every rank afterwards
    """

    def __init__(self, model, interactions, rank, world, device=0, dist=None):
        from ._lightfm_fast import CSRMatrix
        from .lightfm import _Session
        self.model, self.rank, self.world, self.dist = model, rank, world, dist
        shard, self.bounds = local_shard(interactions, rank, world)
        self.shard = shard
        n_users, n_items = shard.shape
        if model.item_embeddings is None:
            model._initialize(model.no_components, n_items, n_users)
        if world > 1:  # replicas start from rank 0's tables
            import torch
            for name in ("item_embeddings", "user_embeddings"):
                dist.broadcast(torch.from_numpy(getattr(model, name)), src=0)
        user_f = sp.identity(n_users, dtype=np.float32, format="csr")
        item_f = sp.identity(n_items, dtype=np.float32, format="csr")
        self.positives = model._get_positives_lookup_matrix(shard)
        self.struct = model._get_lightfm_data()
        self.session = _Session(self.struct, CSRMatrix(item_f), CSRMatrix(user_f), device=device)
        self.session.set_interactions(CSRMatrix(self.positives), np.ascontiguousarray(shard.row),
                                      np.ascontiguousarray(shard.col), shard.data, shard.data)
        if world > 1:
            import torch
            uid = C.create_string_buffer(N.UNIQUE_ID_BYTES)
            if rank == 0:
                N.check(N.lib().lfm_comm_unique_id(uid))
            t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).clone()
            dist.broadcast(t, src=0)
            self.session.comm_init(C.create_string_buffer(bytes(t.numpy().tobytes()),
                                                          N.UNIQUE_ID_BYTES), rank, world)

    def run(self, epochs, num_threads=1):
        from ._lightfm_fast import make_opts
        m = self.model
        n = self.shard.nnz
        stats = []
        for epoch in range(epochs):
            shuffle = np.arange(n, dtype=np.int32)
            m.random_state.shuffle(shuffle)
            seeds = np.ascontiguousarray(m.random_state.randint(
                0, np.iinfo(np.int32).max, size=num_threads).astype(np.uint32))
            self.session.upload_shuffle(shuffle)
            opts, _ = make_opts()
            opts.history = int(getattr(m, "_trained_interactions", 0))
            m._trained_interactions = opts.history + n
            self.session.epoch(m.loss, m.item_alpha, m.user_alpha, m.k, m.n, seeds, opts)
            stats.append(opts)
            if not self.session.check_finite():
                raise ValueError("Not all estimated parameters are finite")
        self.session.comm_merge_users()
        self.session.sync_to_host(self.struct)
        return stats

    def close(self):
        self.session.close()
