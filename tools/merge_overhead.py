"""What a merge of the replicated item tables costs on ONE GPU beyond its bytes on the wire: a one-rank RCCL communicator
(the all-reduce is then a device copy) at the ML-20M shape (26 744 item rows, d = 64: 13.9 MB of packed deltas), wall time
of lfm_session_comm_merge_sparse, synchronous, after a training segment of 1/8 of the interactions (what one of 8 ranks
trains between merges).  Three paths: the dense merge of round 2, the detected-rows sparse merge (fraction 2.0: never all
rows), the all-rows sparse merge (fraction 0).  usage: python tools/merge_overhead.py [reps]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from lightfm_amd import LightFM, _native as N  # noqa: E402
from lightfm_amd import synthetic  # noqa: E402
from lightfm_amd._lightfm_fast import CSRMatrix, make_opts  # noqa: E402
from lightfm_amd.distributed import local_shard  # noqa: E402
from lightfm_amd.lightfm import _Session  # noqa: E402
import scipy.sparse as sp  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
nu, ni, nnz = synthetic.SHAPES["ml-20m"]
data = synthetic.make_interactions(nu, ni, nnz, seed=42)
shard, bounds = local_shard(data, 0, 8, rebase=True)
nu_l = shard.shape[0]
m = LightFM(no_components=64, loss="warp", random_state=3)
m._initialize(64, ni, nu_l)
st = m._get_lightfm_data()
s = _Session(st, CSRMatrix(sp.identity(ni, dtype=np.float32, format="csr")), CSRMatrix(sp.identity(nu_l, dtype=np.float32, format="csr")))
s.set_interactions(None, np.ascontiguousarray(shard.row), np.ascontiguousarray(shard.col), shard.data, shard.data)
s.build_positives(nu_l, ni)
uid = C.create_string_buffer(N.UNIQUE_ID_BYTES)
N.check(N.lib().lfm_comm_unique_id(uid))
s.comm_init(uid, 0, 1)
n = shard.nnz
out = {}
for name, frac in (("dense", None), ("sparse_detect", 2.0), ("sparse_all_rows", 0.0)):
    if frac is not None:
        s.set_merge_dense_fraction(frac)
    times, train = [], []
    for e in range(reps):
        s.device_shuffle(7 + e, 3)
        o, _ = make_opts()
        o.history = 1 << 40
        o.pos_begin, o.pos_end = 0, n // 2   # ~1.25 M interactions: the interval between two merges of an 8-rank job
        t0 = time.perf_counter()
        s.epoch("warp", 0.0, 0.0, 5, 10, np.array([11 + e], np.uint32), o)
        t1 = time.perf_counter()
        if name == "dense":
            s.comm_merge(1, N.MERGE_ADAGRAD)
        else:
            s.comm_merge_sparse(1, N.MERGE_ADAGRAD, overlap=False)
        t2 = time.perf_counter()
        train.append(t1 - t0)
        times.append(t2 - t1)
    out[name] = (1e3 * float(np.median(times[2:])), 1e3 * float(np.min(times[2:])), 1e3 * float(np.median(train[2:])))
    print("%-16s merge call: median %.3f ms, min %.3f ms   (segment of %d interactions: %.3f ms)" % (
        name, out[name][0], out[name][1], n // 2, out[name][2]), flush=True)
s.close()
