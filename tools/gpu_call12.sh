#!/bin/bash
# Round 2, GPU visit 12 (final state): smoke, the full GPU suite, the full bench lines of the four configurations,
# kernel-trace + PMC profiles.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02l
mkdir -p $OUT
cd $R
timeout -k 5 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
AMD_LOG_LEVEL=1 LD_PRELOAD=$R/tools/_bin/libaborttrace.so timeout -k 5 1100 python -m pytest tests -m gpu -q -rP --durations=8 -p no:cacheprovider > $OUT/pytest_full.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_full.log
grep -v "^  File\|Extension modules\|Unknown Event Type" $OUT/pytest_full.log | grep -E "abort_trace|\.so\(|delta|FAILED|ERROR|passed|failed|s call|pytest exit|rocdevice|Fatal" | cut -c1-250 | head -60
line() { python - "$1" <<PY
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]; q = d.get("quality") or {}; c = d.get("cpu_baseline") or {}; f = d.get("end_to_end_fit") or {}
    print("%s: %.1f M/s frac %.3f atomic %.3f in_flight %d launch %.2f ms | p@10 %s ref %s | cpu %s | e2e %s" % (d["config"]["name"], d["value"]/1e6, r["frac"], r["atomic_unit"]["frac"], r["interactions_in_flight"], r["avg_launch_ms"], q.get("precision_at_10"), q.get("precision_at_10_ref"), c.get("value"), f.get("value")))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
timeout -k 5 600 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; line $OUT/bench_c2.json
timeout -k 5 600 python bench.py --config c3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; line $OUT/bench_c3.json
timeout -k 5 600 python bench.py --config c4shard > $OUT/bench_c4shard.json 2> $OUT/bench_c4shard.err; line $OUT/bench_c4shard.json
timeout -k 5 600 python bench.py --config c5shard --scale 0.25 --steps 5 > $OUT/bench_c5shard.json 2> $OUT/bench_c5shard.err; line $OUT/bench_c5shard.json
bash tools/profile2.sh r02l_c2 --config c2
bash tools/profile2.sh r02l_c4shard --config c4shard
bash tools/profile2.sh r02l_c3 --config c3
TRACE_ONLY=1 bash tools/profile2.sh r02l_c5shard --config c5shard --scale 0.25
du -sh $R/gpurun_out | tail -1
