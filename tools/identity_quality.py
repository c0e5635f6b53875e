"""precision@10 of an identity-feature BPR / logistic model against the reference's OpenMP build, N seeds per arm, on synthetic data of a
given shape (the precision gate's generator, tests/test_precision_parity.py): the shipped kernel plan, the row-stream kernels
(LIGHTFM_AMD_BPR_TILE=0 LIGHTFM_AMD_LOGISTIC_TILE=0 LIGHTFM_AMD_BPR_WIDE_TILE=0) and the shipped plan at 64 wavefronts.
    python tools/identity_quality.py [loss=bpr] [d=10] [n_users=8656] [n_items=6686] [nnz=1000000] [n_seeds=32] [epochs=5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from tests.test_precision_parity import _data, _p10, _with_explicit_negatives
from lightfm_amd import LightFM, options
from oracle.ref_model import RefLightFM

arg = lambda i, default, cast=int: cast(sys.argv[i]) if len(sys.argv) > i else default
loss, d = arg(1, "bpr", str), arg(2, 10)
nu, ni, nnz, n_seeds, epochs = arg(3, 8656), arg(4, 6686), arg(5, 1_000_000), arg(6, 32), arg(7, 5)
train, test = _data(nu, ni, nnz)
fit_on = _with_explicit_negatives(train, test) if loss == "logistic" else train
tr, te = train.tocsr(), test.tocsr()
seeds = list(range(1, n_seeds + 1))
se = lambda x: float(np.std(x, ddof=1) / np.sqrt(len(x)))
print("# %s d=%d on %d x %d, %d train interactions, %d epochs, %d seeds per arm (seeds 1..%d on every arm), mean +- standard error" % (loss, d, nu, ni, train.nnz, epochs, n_seeds, n_seeds))

def fit_ref(seed):
    r = RefLightFM(no_components=d, loss=loss, random_state=seed)
    r.fit(fit_on, epochs=epochs, num_threads=min(16, os.cpu_count() or 1))
    return r

res = {}
with ThreadPoolExecutor(max_workers=3) as pool:
    pending = [pool.submit(fit_ref, s) for s in seeds]
    for arm, env, waves in (("shipped plan", "1", 0), ("row-stream kernels", "0", 0), ("shipped plan, max_waves 64", "1", 64)):
        for k in ("LIGHTFM_AMD_BPR_TILE", "LIGHTFM_AMD_LOGISTIC_TILE", "LIGHTFM_AMD_BPR_WIDE_TILE"):
            os.environ[k] = env
        options.set(max_waves=waves)
        out = []
        for s in seeds:
            m = LightFM(no_components=d, loss=loss, random_state=s)
            m.fit(fit_on, epochs=epochs)
            out.append(_p10(m, tr, te, None))
        st = m._last_epoch_stats[-1]
        res[arm] = out
        print("%-26s %.4f +- %.4f   (kernel_used %d, plan flags %d, in flight %d)" % (arm, np.mean(out), se(out), st["kernel_used"], st["plan_flags"], st["in_flight"]), flush=True)
    ref = [_p10(f.result(), tr, te, None) for f in pending]
print("%-26s %.4f +- %.4f" % ("reference, 16 threads", np.mean(ref), se(ref)))
for arm, out in res.items():
    print("delta %-26s %+.4f +- %.4f" % (arm, np.mean(out) - np.mean(ref), float(np.hypot(se(out), se(ref)))))
