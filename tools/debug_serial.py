import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from tests import helpers as H
from tests.test_oracle_vs_reference import LOSS_CASES, _problem, _run_orc
from tests.test_hip_parity import _run_hip, _orc_logged
import lightfm_amd._lightfm_fast as fast
from lightfm_amd.options import options

def diff(a, b):
    for n in oracle.ARRAYS:
        x, y = getattr(a, n), getattr(b, n)
        nd = int((x != y).sum())
        if nd:
            print("   %-28s ndiff %6d / %6d  maxabs %.3g" % (n, nd, x.size, np.abs(x - y).max()))

for loss in ("logistic", "warp"):
    for nlim in (1, 2, 5, 50, None):
        options.set(mode="serial", log_samples=True)
        coo, item_f, user_f, st, rng, alpha = _problem(LOSS_CASES[0])
        if nlim:
            import scipy.sparse as sp
            coo = sp.coo_matrix((coo.data[:nlim], (coo.row[:nlim], coo.col[:nlim])), shape=coo.shape)
        a, b = st.copy(), st.copy()
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _run_hip(fast, loss, coo, item_f, user_f, a, shuffle, seeds, 0.0)
        print(loss, "n=", len(coo.data), "counters", options.last_counters, "ms", options.last_kernel_ms)
        if loss == "logistic":
            _run_orc(loss, coo, item_f, user_f, b, shuffle, seeds, 0.0)
        else:
            o = _orc_logged(loss, coo, item_f, user_f, b, shuffle, seeds, 0.0, rng_mode=0)
            neg, sampled = options.last_logs
            print("   sampled equal", np.array_equal(sampled, o.sampled), "neg equal", np.array_equal(neg, o.neg), "orc counters", o.counters)
            if not np.array_equal(sampled, o.sampled):
                i = int(np.argmax(sampled != o.sampled)); print("   first diff at", i, sampled[max(0,i-2):i+3], o.sampled[max(0,i-2):i+3])
        diff(a, b)
        diff_init = sum(int((getattr(a, n) != getattr(st, n)).sum()) for n in oracle.ARRAYS)
        print("   cells changed vs init (hip):", diff_init, " (orc):", sum(int((getattr(b, n) != getattr(st, n)).sum()) for n in oracle.ARRAYS))
