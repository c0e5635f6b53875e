#!/bin/bash
# Round 2, GPU visit 10: speculative first-level in_positives pivots in the tile kernel (tests + bench).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02j
mkdir -p $OUT
cd $R
LD_PRELOAD=$R/tools/_bin/libaborttrace.so timeout -k 5 400 python -m pytest tests/test_hip_warp_tile.py tests/test_hip_parity.py -q -x -p no:cacheprovider > $OUT/pytest_tile.log 2>&1; echo "tile tests exit $?"; tail -4 $OUT/pytest_tile.log | cut -c1-300
Q="--no-cpu-baseline --no-quality --no-fit"
run() { cfg=$1; tag=$2; shift; shift; timeout -k 5 300 env "$@" python bench.py --config $cfg $Q $EXTRA > $OUT/${cfg}_$tag.json 2> $OUT/${cfg}_$tag.err; python - <<PY
import json
try:
    d = json.load(open("$OUT/${cfg}_$tag.json")); r = d["roofline"]
    print("%s %-14s %8.1f M/s frac %.3f in_flight %5d launch %6.2f ms eps %d k/step %.2f S %.2f U %.2f %s" % ("$cfg", "$tag", d["value"]/1e6, r["frac"], r["interactions_in_flight"], r["avg_launch_ms"], d["config"]["epochs_per_step"], r["kernel_time_fraction_of_step"], r["draws_per_interaction"], r["updates_per_interaction"], r.get("phase_cycles_per_pass") or r.get("phase_cycles_per_interaction") or ""))
except Exception as e:
    print("$cfg $tag FAILED", e); import subprocess; print(subprocess.run("tail -3 $OUT/${cfg}_$tag.err", shell=True, capture_output=True, text=True).stdout)
PY
}
E8="--steps 3 --warmup 1 --epochs-per-step 8"
EXTRA="$E8" run c2 pivots A=1
EXTRA="$E8 --warp-kernel 2" run c2 pivots_timed A=1
EXTRA="$E8 --debug 64" run c2 pivots_regs A=1
EXTRA="--steps 2 --warmup 1 --epochs-per-step 2" run c4shard pivots A=1
