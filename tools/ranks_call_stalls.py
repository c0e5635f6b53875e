"""Wall and CPU time of repeated predict_rank calls at the ML-20M shape next to the container's CPU quota and throttle counters
(the round-5 hunt of the 75 ms stalls of every other call: a BLAS probe in the scoring-session signature woke all 256 cores and
ran the process into its 16-CPU cgroup quota; profiles/r05_visit_k.txt)."""
import sys, time, os
sys.path.insert(0, ".")
import numpy as np
from lightfm_amd import LightFM, synthetic, _native as N
def cg():
    out = {}
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.stat"):
        try: out[f] = open(f).read().replace("\n", " ")
        except Exception as e: pass
    return out
print(cg(), "cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
data = synthetic.make_interactions(138493, 26744, 21000000, seed=42)
train, test = synthetic.split_off_test(data, 20000263, seed=1)
m = LightFM(no_components=64, loss="warp", random_state=3).fit(train, epochs=1)
tc, trc = test.tocsr().astype(np.float32), train.tocsr().astype(np.float32)
a = cg()
w = []
for i in range(16):
    t = time.perf_counter(); c0 = time.process_time(); r = m.predict_rank(tc, train_interactions=trc, check_intersections=False)
    w.append((1e3 * (time.perf_counter() - t), 1e3 * (time.process_time() - c0)))
print("wall/cpu per call:", " ".join("%.0f/%.0f" % x for x in w), flush=True)
print(a); print(cg())
