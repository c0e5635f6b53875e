"""Does RCCL initialise in a process that has imported torch (bench.py / DistributedFit at N > 1 use torch.distributed's gloo backend
for the rendezvous)?  MODE=late: torch first, RCCL at comm_init (what round 4 shipped); MODE=early: librccl loaded (a unique id drawn)
before torch is imported.  One rank, one GPU."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

mode = os.environ.get("MODE", "late")
from lightfm_amd import LightFM, _native as N
assert N.device_count() > 0
uid = C.create_string_buffer(N.UNIQUE_ID_BYTES)
if mode == "early":
    N.check(N.lib().lfm_comm_unique_id(uid))   # loads librccl now
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29566")
dist.init_process_group(backend="gloo", rank=0, world_size=1)
from lightfm_amd._lightfm_fast import CSRMatrix
from lightfm_amd.lightfm import _Session
import scipy.sparse as sp
m = LightFM(no_components=16, loss="warp", random_state=1); m._initialize(16, 50, 40)
s = _Session(m._get_lightfm_data(), CSRMatrix(sp.identity(50, dtype=np.float32, format="csr")), CSRMatrix(sp.identity(40, dtype=np.float32, format="csr")))
try:
    if mode != "early":
        N.check(N.lib().lfm_comm_unique_id(uid))
    s.comm_init(uid, 0, 1)
    s.comm_barrier()
    print("MODE=%s: RCCL communicator initialised in a process with torch %s loaded" % (mode, torch.__version__), flush=True)
except Exception as e:
    print("MODE=%s: FAILED: %r" % (mode, e), flush=True)
s.close()
import subprocess
print(subprocess.run("grep -E 'hip|hsa|rccl' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid(), shell=True, capture_output=True, text=True).stdout)
