"""Turn rocprofv3's rocpd sqlite outputs (gpurun_out/prof_<tag>/...) into the small
text/JSON summaries committed under profiles/.

    python tools/prof_summary.py <tag>
"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return con.execute(sql).fetchall()
    finally:
        con.close()


lines = []
trace = os.path.join(src, "trace", "trace_results.db")
if os.path.exists(trace):
    rows = q(trace, "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                    "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                    "from kernels group by name order by sum(duration) desc")
    tot = sum(r[2] for r in rows) or 1
    lines.append("# rocprofv3 --kernel-trace --stats -- python bench.py   (tag %s)" % tag)
    lines.append("%-60s %6s %12s %12s %12s %12s %6s %5s %5s %7s %9s" % (
        "kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds_B", "grid"))
    for r in rows:
        lines.append("%-60s %6d %12.3f %12.1f %12.1f %12.1f %6.2f %5d %5d %7d %9d" % (
            r[0][:60], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot,
            r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0))
    # Consecutive launches of the epoch kernel run on two streams and overlap by the draining tail of the earlier
    # one (csrc/session.hip): a launch's own duration then includes the time its first workgroups wait for slots.
    # The time the chip actually spends per launch is the UNION of the launches' intervals / their number.
    try:
        con = sqlite3.connect(trace)
        cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
        c0 = "start" if "start" in cols else [c for c in cols if "start" in c][0]
        c1 = "end" if "end" in cols else [c for c in cols if c.startswith("end")][0]
        top = rows[0][0]
        iv = sorted(con.execute('select "%s", "%s" from kernels where name = ?' % (c0, c1), (top,)).fetchall())
        con.close()
        union, cur_s, cur_e = 0, None, None
        for a, b in iv:
            if cur_e is None or a > cur_e:
                if cur_e is not None:
                    union += cur_e - cur_s
                cur_s, cur_e = a, b
            else:
                cur_e = max(cur_e, b)
        if cur_e is not None:
            union += cur_e - cur_s
        lines.append("")
        lines.append("# %s: %d launches, sum of their durations %.3f ms, UNION of their intervals %.3f ms = %.1f us per launch "
                     "(launches overlap by %.1f %%: two streams)" % (top[:60], len(iv), rows[0][2] / 1e6, union / 1e6,
                                                                    union / 1e3 / max(1, len(iv)),
                                                                    100.0 * (rows[0][2] - union) / max(1, rows[0][2])))
    except Exception as e:  # schema differences: the per-kernel table above stands on its own
        lines.append("# (union of launch intervals not available: %r)" % (e,))
    hs = [r for r in rows if "hot_slice_kernel" in r[0]]
    ff = [r for r in rows if "fit_feat_kernel" in r[0]]
    if hs and ff:
        # the hot set (csrc/hot_slices.hip): a launch of the row-stream kernel and the slice kernel that applies its records run
        # one after the other on one stream -- the PAIR (with the record memset and the snapshot) is what a "launch" of
        # bench.py's roofline is
        extra = [r for r in rows if "hot_snapshot_kernel" in r[0] or "fillBuffer" in r[0]]
        n = ff[0][1]
        tot_ns = ff[0][2] + hs[0][2] + sum(r[2] for r in extra)
        lines.append("# hot set: %d launches of (row-stream kernel %.1f us + slice kernel %.1f us + memset / snapshot %.1f us) = %.1f us per "
                     "launch pair ON AVERAGE -- the launch length ramps with the training history (clamp(history / 128, 8, 128 Ki) positions), "
                     "so this mean over short and long launches is NOT the duration of the full-length launch the bench line's "
                     "algorithmic_bytes_per_launch refers to" % (n, ff[0][3] / 1e3, hs[0][3] / 1e3, sum(r[2] for r in extra) / 1e3 / max(1, n), tot_ns / 1e3 / max(1, n)))
        try:
            b = json.load(open(os.path.join(src, "bench_trace.json")))
            r = b["roofline"]
            per_epoch = r["algorithmic_bytes_per_launch"] / r["algorithmic_bytes_per_interaction"] * r["launches_per_epoch"]
            epochs = b["steps"] * b["config"].get("epochs_per_step", 1) + b["warmup"] * b["config"].get("epochs_per_step", 1)
            total_bytes = per_epoch * epochs * r["algorithmic_bytes_per_interaction"]
            lines.append("# consistent pair: ALL %d profiled epochs (%.1f M interactions x %.0f B algorithmic = %.3f TB) / ALL these kernels (%.1f ms) "
                         "= %.3f TB/s = %.3f of 8 TB/s, the first epoch's ramp and the profiler's overhead included; the timed epochs alone: "
                         "roofline.frac of the line below" % (epochs, per_epoch * epochs / 1e6, r["algorithmic_bytes_per_interaction"], total_bytes / 1e12,
                                                             tot_ns / 1e6, total_bytes / tot_ns / 1e3, total_bytes / tot_ns / 1e3 / 8.0))
        except Exception as e:
            lines.append("# (totals not available: %r)" % (e,))
    bj = os.path.join(src, "bench_trace.json")
    if os.path.exists(bj):
        lines.append("")
        lines.append("# bench.py JSON line of the same (profiled) run:")
        lines.append(open(bj).read().strip())
    open(os.path.join(dst, "%s_kernel_stats.txt" % tag), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))

pmc = {}
for sub in sorted(os.listdir(src)):
    db = os.path.join(src, sub, "pmc_results.db")
    if not os.path.exists(db):
        continue
    rows = q(db, "select k.name, p.counter_name, count(*), sum(p.counter_value) from pmc_events p "
                 "join kernels k on k.dispatch_id = p.dispatch_id "
                 "group by k.name, p.counter_name order by k.name")
    for name, cname, cnt, val in rows:
        if "fit_" not in name and "hot_slice_kernel" not in name:
            continue
        pmc.setdefault(name, {})[cname] = {"dispatches": cnt, "sum": val, "per_launch": val / cnt}
summary = {"tag": tag, "command": "rocprofv3 --kernel-trace --pmc <counter(s)> -- python bench.py --steps 2 "
                                  "--warmup 1 --no-cpu-baseline (one pass per counter group)",
           "kernels": pmc}
# HBM traffic of the DOMINANT kernel (the epoch also has a few short launches of the other tile
# variants while the concurrency ramp is below the chip's residency)
dominant = max((n for n in pmc if "fit_" in n and "FETCH_SIZE" in pmc[n] and "WRITE_SIZE" in pmc[n]),
               key=lambda n: pmc[n]["FETCH_SIZE"]["sum"], default=None)
summary["dominant_kernel"] = dominant
# the hot set (round 6): every launch of the row-stream kernel is followed by one hot_slice_kernel launch that applies its
# records -- the pair is the unit bench.py times; the slice kernel's bytes are added to the dominant kernel's per launch
companions = [n for n in pmc if "hot_slice_kernel" in n and "FETCH_SIZE" in pmc[n] and "WRITE_SIZE" in pmc[n]]
summary["companion_kernels"] = companions
for name, c in pmc.items():
    if name == dominant:
        # FETCH_SIZE / WRITE_SIZE are reported in KiB (rocprofv3); on gfx950 FETCH_SIZE counts
        # 128-B requests at 64 B (MI355X_MICROARCH.md "HBM"): the x2 correction applies to wide
        # coalesced streams; this kernel's reads are 4 B/lane row gathers (256-B rows), so both
        # the raw and the x2-corrected figures are recorded.
        f, w = c["FETCH_SIZE"]["per_launch"] * 1024.0, c["WRITE_SIZE"]["per_launch"] * 1024.0
        for comp in companions:
            f += pmc[comp]["FETCH_SIZE"]["per_launch"] * 1024.0
            w += pmc[comp]["WRITE_SIZE"]["per_launch"] * 1024.0
        summary["hbm_bytes_per_launch_raw"] = f + w
        summary["hbm_bytes_per_launch"] = 2 * f + w
        summary["fetch_bytes_per_launch_raw"] = f
        summary["write_bytes_per_launch"] = w
        # Calibrated by request size (round 5): FETCH_SIZE tallies EVERY memory-side read request at 64 B, so doubling it is
        # right for 128-byte requests only.  With the request-size counters of the same command: read bytes = 128 n_128 +
        # 32 n_32 + 64 (n - n_128 - n_32); the write side keeps WRITE_SIZE (64-byte / 32-byte write requests; a float
        # atomic travels as a write request).
        if all(k in c for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_128B_sum")):
            n = c["TCC_EA0_RDREQ_sum"]["per_launch"]
            n32, n128 = c["TCC_EA0_RDREQ_32B_sum"]["per_launch"], c["TCC_EA0_RDREQ_128B_sum"]["per_launch"]
            for comp in companions:
                if all(k in pmc[comp] for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_128B_sum")):
                    n += pmc[comp]["TCC_EA0_RDREQ_sum"]["per_launch"]
                    n32 += pmc[comp]["TCC_EA0_RDREQ_32B_sum"]["per_launch"]
                    n128 += pmc[comp]["TCC_EA0_RDREQ_128B_sum"]["per_launch"]
            rd = 128.0 * n128 + 32.0 * n32 + 64.0 * max(0.0, n - n128 - n32)
            summary["read_requests_per_launch"] = {"all": n, "32B": n32, "128B": n128, "64B": max(0.0, n - n128 - n32)}
            summary["read_bytes_per_launch_calibrated"] = rd
            summary["hbm_bytes_per_launch_calibrated"] = rd + w
            if "TCC_EA0_WRREQ_sum" in c and "TCC_EA0_WRREQ_64B_sum" in c:
                nw, nw64 = c["TCC_EA0_WRREQ_sum"]["per_launch"], c["TCC_EA0_WRREQ_64B_sum"]["per_launch"]
                summary["write_requests_per_launch"] = {"all": nw, "64B": nw64, "atomic": c.get("TCC_EA0_ATOMIC_sum", {}).get("per_launch")}
                summary["write_bytes_per_launch_by_requests"] = 64.0 * nw64 + 32.0 * max(0.0, nw - nw64)
json.dump(summary, open(os.path.join(dst, "%s_pmc_summary.json" % tag), "w"), indent=1, sort_keys=True)
json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True)[:3000])
