"""Can RCCL form a communicator of several ranks on ONE GPU?  (The box has one GPU; a yes would let the merge path run with
real ranks.)  RANK / WORLD_SIZE / MASTER_* from the environment; every rank uses device 0.

    for r in 0 1; do RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 python tools/rccl_two_ranks_one_gpu.py & done; wait
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
from lightfm_amd import LightFM, _native as N, synthetic
assert N.device_count() > 0
import torch.distributed as dist
from lightfm_amd.distributed import DistributedFit

dist.init_process_group(backend="gloo", rank=rank, world_size=world)
data = synthetic.make_interactions(4000, 3000, 400000, seed=3)
model = LightFM(no_components=32, loss="warp", random_state=5)
try:
    fit = DistributedFit(model, data, rank, world, device=0, dist=dist)
    stats = fit.run(3)
    print("rank %d: communicator of %d ranks on one GPU WORKS: %d merges, %.1f KB per merge, item table sum %.6f" % (
        rank, world, fit.merges, fit.merge_bytes / 1e3 / max(1, fit.merges), float(model.item_embeddings.astype(np.float64).sum())), flush=True)
    fit.close()
except Exception as e:
    print("rank %d: FAILED: %r" % (rank, e), flush=True)
dist.destroy_process_group()
