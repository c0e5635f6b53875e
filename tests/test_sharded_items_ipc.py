"""Owner-sharded item tables over PROCESSES (include/lfm_hip.h: lfm_session_share_items_ipc): K processes on the ONE
GPU of the box export / map each other's item-side allocations through HIP IPC and train against the owners' rows --
the multi-process form of tests/test_sharded_items.py, and what runs one process per GPU (peer mappings over xGMI) on
a multi-GPU node.  The ranks (tests/ipc_worker.py) check: frozen-weight samples equal the oracle's on the true model;
after concurrent training no rank wrote a row it does not own (poisoned copies intact) while every owner's rows were
trained by all ranks; after lfm_session_gather_shared_items all ranks hold the same complete, finite item tables;
lightfm_amd.distributed.DistributedFit(item_tables="owner") trains a model every rank agrees on."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 4])
def test_processes_share_owner_sharded_item_tables(world):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONFAULTHANDLER="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ipc_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=420)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "IPC_WORKER_OK rank %d of %d" % (rank, world) in out, \
            "rank %d exited %s:\n%s" % (rank, p.returncode, out[-3000:])
