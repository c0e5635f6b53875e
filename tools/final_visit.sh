#!/bin/bash
# tools/final_visit.sh -- after the width condition on the plain-store user rows (csrc/session.hip): the GPU suite without the precision-gate
# file (its problems are too small for the switch either way), smoke, the plans / rates of the widths the condition touches, the driver's bench
cd ${GRAFT_REPO_ROOT:-/root/repo}; OUT=gpurun_out/r6zz; mkdir -p $OUT
( time timeout 900 python3 -m pytest -p no:cacheprovider tests -m gpu -x -q --deselect tests/test_precision_parity.py ) > $OUT/suite.txt 2>&1; tail -4 $OUT/suite.txt | cut -c1-300
timeout 300 python3 __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 300 python3 tools/width_sweep.py warp,bpr,logistic 16,20,32,48,64 > $OUT/width.txt 2>&1; cat $OUT/width.txt | cut -c1-250
( time timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; echo "bench bytes $(wc -c < $OUT/bench.json)"
python3 - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("c2 %.1f M/s frac %.3f  traffic %s" % (d["value"] / 1e6, d["roofline"]["frac"], d["roofline"].get("traffic_over_algorithmic")))
for k, v in d["config"].get("legs", {}).items():
    try: print("  %-10s %s  frac %s" % (k, v.get("value"), (v.get("roofline") or {}).get("frac")))
    except Exception as e: print(k, e)
print("  fit", (d["config"].get("end_to_end_fit") or {}).get("value"), "quality", d["config"].get("quality"))
PY
