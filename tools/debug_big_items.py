"""Diagnostic of tests/test_hbm_shapes.py::test_item_table_beyond_4_gb...: which cells of which table change in a
frozen-weight epoch when the item table is beyond 4 GB.   python tools/debug_big_items.py [n_items] [debug bits]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from lightfm_amd import synthetic
from lightfm_amd._lightfm_fast import CSRMatrix, FastLightFM, make_opts
from lightfm_amd.lightfm import _Session
from lightfm_amd.options import options
from oracle import oracle
from tests import helpers as H
from tests.test_hbm_shapes import _big_state

ni = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
debug = int(sys.argv[2]) if len(sys.argv) > 2 else 0
epoch = (sys.argv[3] != "noepoch") if len(sys.argv) > 3 else True
nu, d = 300_000, 64
coo = synthetic.big_interactions(nu, ni, 2_500_000, seed=6)
n = coo.nnz
sc = 3.0 / d ** 0.25
st = _big_state(ni, nu, d, 37, sc, sc)
small = {k: getattr(st, k).copy() for k in oracle.ARRAYS if "embedding" not in k}
zeros = np.zeros_like(coo.data)
seeds = np.array([20240917], np.uint32)
item_f, user_f = H.identity_features(ni), H.identity_features(nu)
options.set(mode="parallel", launches_per_epoch=0, ramp_k=-1, debug=debug)
fl = FastLightFM(*st.arrays(), d, 0, st.lr, st.rho, st.eps, st.max_sampled)
session = _Session(fl, CSRMatrix(item_f), CSRMatrix(user_f))
session.set_interactions(None, np.ascontiguousarray(coo.row), np.ascontiguousarray(coo.col), coo.data, zeros)
session.build_positives(nu, ni)
session.device_shuffle(97531, 86420)
opts, logs = make_opts(n, want_log=True)
if epoch:
    session.epoch("warp", 0.0, 0.0, 5, 10, seeds, opts)
session.sync_to_host(fl)
session.close()
print("ni %d debug %d epoch %s: kernel %d ng %d ahead %d flags %d" % (ni, debug, epoch, opts.kernel_used, opts.tile_ng, opts.tile_ahead, opts.plan_flags))
for k, old in small.items():
    new = getattr(st, k)
    bad = np.flatnonzero(new.view(np.uint32) != old.view(np.uint32))
    print("%-24s changed %d of %d" % (k, len(bad), old.size), end="")
    if len(bad):
        print("  index range [%d, %d]  first: %s" % (bad[0], bad[-1], [(int(i), float(old[i]), float(new[i])) for i in bad[:6]]), end="")
    print()
