"""profiles/traffic.json from the PMC summaries of a round's profile runs.

    python tools/update_traffic.py r04 c2 c3 c4shard c5shard

For every config: the dominant epoch kernel's HBM bytes per launch (profiles/<round>_<cfg>_pmc_summary.json,
written by tools/prof_summary.py from the rocprofv3 --pmc passes of tools/profile2.sh) AND per interaction --
divided by the interactions per launch of the SAME profiled run (gpurun_out/prof_<round>_<cfg>/bench_fetch.json),
next to that run's own algorithmic bytes, updates and draws per interaction: bench.py scales the per-interaction
figure to its own launch length, and the traffic / algorithmic ratio it prints is the profiled run's own pair
(the counters cannot be collected inside a bench run; the two runs differ in the epochs they cover).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, cfgs = sys.argv[1], sys.argv[2:]
path = os.path.join(ROOT, "profiles", "traffic.json")
table = json.load(open(path)) if os.path.exists(path) else {}
table["_about"] = ("HBM traffic of the dominant epoch kernel from the rocprofv3 --pmc passes of tools/profile2.sh (separate runs per "
                   "counter group; 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md as "
                   "applied by tools/prof_summary.py), per launch and per interaction of the PROFILED run, with that run's own "
                   "algorithmic bytes / updates / draws per interaction.  bench.py copies it into roofline.traffic when its run used "
                   "the same kernel (tools/update_traffic.py).")
for cfg in cfgs:
    tag = "%s_%s" % (rnd, cfg)
    summ = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_summary.json")))
    entry = {"kernel": summ["dominant_kernel"], "hbm_bytes_per_launch": summ["hbm_bytes_per_launch"],
             "source": "profiles/%s_pmc_summary.json" % tag, "command": "tools/profile2.sh %s --config %s" % (tag, cfg)}
    if "hbm_bytes_per_launch_calibrated" in summ:
        # the read side by request size (128 / 64 / 32-byte memory-side requests) instead of 2 x FETCH_SIZE: the figure
        # bench.py reports; the uncalibrated one is kept beside it
        entry["hbm_bytes_per_launch_2x_fetch"] = summ["hbm_bytes_per_launch"]
        entry["hbm_bytes_per_launch"] = summ["hbm_bytes_per_launch_calibrated"]
        entry["calibration"] = "read bytes = 128 n_128B + 64 n_64B + 32 n_32B (TCC_EA0_RDREQ*), writes = WRITE_SIZE"
        entry["read_requests_per_launch"] = summ.get("read_requests_per_launch")
        summ = dict(summ, hbm_bytes_per_launch=summ["hbm_bytes_per_launch_calibrated"])
    for leg in ("bench_fetch.json", "bench_write.json", "bench_trace.json"):
        bj = os.path.join(ROOT, "gpurun_out", "prof_" + tag, leg)
        if os.path.exists(bj) and os.path.getsize(bj) > 10:
            b = json.load(open(bj))
            r = b["roofline"]
            per_launch = r["algorithmic_bytes_per_launch"] / r["algorithmic_bytes_per_interaction"]
            entry.update({"interactions_per_launch_profiled": per_launch,
                          "hbm_bytes_per_interaction": summ["hbm_bytes_per_launch"] / per_launch,
                          "algorithmic_bytes_per_interaction_profiled": r["algorithmic_bytes_per_interaction"],
                          "traffic_over_algorithmic_profiled": summ["hbm_bytes_per_launch"] / r["algorithmic_bytes_per_launch"],
                          "updates_per_interaction_profiled": r["updates_per_interaction"],
                          "draws_per_interaction_profiled": r["draws_per_interaction"],
                          "profiled_epochs": b["config"].get("timed_epochs"), "profiled_scale": b["config"].get("scale", 1.0)})
            break
    table[cfg] = entry
    print(cfg, json.dumps(entry)[:400])
json.dump(table, open(path, "w"), indent=1)
