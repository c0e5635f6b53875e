"""precision@10 at the FULL ML-20M shape for WARP at the reference's default width (no_components = 10, the narrow-model tile
kernel with one 128-byte line per feature): this backend in its shipped mode and with fewer interactions in flight / the wide
tile kernel, against the reference's OpenMP build (16 threads per fit), all test users, the reference's metric.  The reference
fits run once (in a thread pool, beside the arms); every arm is a process of its own (the knobs are read once per process).
    python tools/narrow_quality20m.py [epochs=3] [seeds=16] [d=10]
NQ_ARMS='[["name", {"ENV": "value"}], ...]' replaces the arms; NQ_REF="mean,se" takes the reference's numbers from an earlier run of the
same seeds instead of fitting it again (LIGHTFM_AMD_DEBUG=4096 / 2048: user rows by atomics / by plain stores, whatever the rule says)."""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ARMS = [("shipped", {}),
        ("2 workgroups per CU", {"LIGHTFM_AMD_NARROW_BLOCKS": "2"}),
        ("1 workgroup per CU", {"LIGHTFM_AMD_NARROW_BLOCKS": "1"}),
        ("wide tile kernel", {"LIGHTFM_AMD_TILE_PAIRS": "0"})]


if os.environ.get("NQ_ARMS"):
    ARMS = [(a, dict(b)) for a, b in json.loads(os.environ["NQ_ARMS"])]


def problem():
    from lightfm_amd import synthetic
    data = synthetic.named("ml-20m")
    train, test = synthetic.train_test_split(data, 0.05, seed=1)
    return data, train, test


def arm(epochs, seeds, d):
    """One arm in this process: prints a JSON line (precision@10 per seed, the last epoch's plan)."""
    from lightfm_amd import LightFM
    from lightfm_amd.evaluation import precision_at_k
    _, train, test = problem()
    tr, te = train.tocsr(), test.tocsr()
    out, t_fit = [], []
    for s in seeds:
        t = time.time()
        m = LightFM(no_components=d, loss="warp", random_state=s)
        m.fit(train, epochs=epochs)
        t_fit.append(time.time() - t)
        out.append(float(precision_at_k(m, te, train_interactions=tr, k=10).mean()))
    st = m._last_epoch_stats[-1]
    print(json.dumps({"p10": out, "fit_s": float(np.median(t_fit)), "kernel_ms": st["kernel_ms"], "in_flight": st["in_flight"],
                      "plan_flags": st["plan_flags"], "user_store": st["user_store"]}), flush=True)


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    d = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    seeds = list(range(1, n_seeds + 1))
    from concurrent.futures import ThreadPoolExecutor
    from lightfm_amd.evaluation import precision_at_k
    from oracle.ref_model import RefLightFM
    data, train, test = problem()
    tr, te = train.tocsr(), test.tocsr()
    print("# WARP d = %d; ML-20M shape %s, %d train / %d test interactions, %d epochs, %d seeds per arm; precision@10 over all %d test users"
          % (d, data.shape, train.nnz, test.nnz, epochs, n_seeds, len(np.unique(test.row))), flush=True)
    se = lambda x: float(np.std(x, ddof=1) / np.sqrt(len(x)))

    def fit_ref(seed):
        r = RefLightFM(no_components=d, loss="warp", random_state=seed)
        r.fit(train, epochs=epochs, num_threads=min(16, os.cpu_count() or 1))
        return r

    workers = max(1, min(n_seeds, (os.cpu_count() or 16) // 16))
    given = os.environ.get("NQ_REF")
    with ThreadPoolExecutor(max_workers=workers) as pool:
        pending = [] if given else [pool.submit(fit_ref, s) for s in seeds]
        results = []
        for name, env in ARMS:
            e = dict(os.environ); e.update(env)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--arm", str(epochs), ",".join(map(str, seeds)), str(d)],
                               env=e, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not line:
                print("  %-22s FAILED rc %d: %s" % (name, p.returncode, p.stderr[-400:]), flush=True)
                continue
            results.append((name, env, json.loads(line[-1])))
        ref = [float(precision_at_k(f.result(), te, train_interactions=tr, k=10).mean()) for f in pending]
    if given:
        ref_mean, ref_se = [float(x) for x in given.split(",")]
        print("  %-22s %.5f +- %.5f   (given: an earlier run of the same seeds)" % ("reference (16 threads)", ref_mean, ref_se), flush=True)
    else:
        ref_mean, ref_se = float(np.mean(ref)), se(ref)
        print("  %-22s %.5f +- %.5f   %s" % ("reference (16 threads)", ref_mean, ref_se, [round(x, 4) for x in ref]), flush=True)
    for name, env, r in results:
        h = r["p10"]
        print("  %-28s %.5f +- %.5f   delta %+.5f +- %.5f   epoch kernels %.2f ms, in flight %d, plan flags %d, user_store %d, fit %.2f s  %s"
              % (name, np.mean(h), se(h), np.mean(h) - ref_mean, np.hypot(se(h), ref_se), r["kernel_ms"], r["in_flight"],
                 r["plan_flags"], r["user_store"], r["fit_s"], env or ""), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--arm":
        arm(int(sys.argv[2]), [int(x) for x in sys.argv[3].split(",")], int(sys.argv[4]))
    else:
        main()
