"""precision@10 of BPR at the reference's default width (no_components = 10), identity features: the lane-group kernel
(csrc/logistic_tile.hip) against the row-stream kernel against the reference's OpenMP build, N seeds per arm on the
precision gate's data (tests/test_precision_parity.py).   python tools/bpr_tile_quality.py [n_seeds=32] [epochs=5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from tests.test_precision_parity import _data, _p10
from lightfm_amd import LightFM
from oracle.ref_model import RefLightFM

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 32
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
train, test = _data(8656, 6686, 1_000_000)
tr, te = train.tocsr(), test.tocsr()
seeds = list(range(1, n_seeds + 1))
se = lambda x: float(np.std(x, ddof=1) / np.sqrt(len(x)))

def fit_ref(seed):
    r = RefLightFM(no_components=10, loss="bpr", random_state=seed)
    r.fit(train, epochs=epochs, num_threads=min(16, os.cpu_count() or 1))
    return r

res = {}
with ThreadPoolExecutor(max_workers=3) as pool:
    pending = [pool.submit(fit_ref, s) for s in seeds]
    for arm, env in (("tile", "1"), ("row-stream", "0"), ("tile, max_waves 64", "1")):
        os.environ["LIGHTFM_AMD_BPR_TILE"] = env
        from lightfm_amd import options
        options.set(max_waves=64 if "max_waves" in arm else 0)
        out = []
        for s in seeds:
            m = LightFM(no_components=10, loss="bpr", random_state=s)
            m.fit(train, epochs=epochs)
            out.append(_p10(m, tr, te, None))
        st = m._last_epoch_stats[-1]
        res[arm] = out
        print("%-20s %.4f +- %.4f   (kernel_used %d, plan flags %d, in flight %d)" % (arm, np.mean(out), se(out), st["kernel_used"], st["plan_flags"], st["in_flight"]), flush=True)
    ref = [_p10(f.result(), tr, te, None) for f in pending]
print("%-20s %.4f +- %.4f" % ("reference, 16 threads", np.mean(ref), se(ref)))
for arm, out in res.items():
    print("delta %-20s %+.4f +- %.4f" % (arm, np.mean(out) - np.mean(ref), float(np.hypot(se(out), se(ref)))))
