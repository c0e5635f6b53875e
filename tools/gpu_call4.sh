#!/bin/bash
# Round 2, GPU visit 4: suite with durations, full bench lines of the four configs.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02d
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -rP --durations=15 -p no:cacheprovider > $OUT/pytest_full.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_full.log
grep -E "delta|passed|failed|^FAILED|^ERROR|s call|s setup" $OUT/pytest_full.log > $OUT/pytest.log
tail -30 $OUT/pytest.log
Q="--no-cpu-baseline --no-quality --no-fit"
run() { cfg=$1; tag=$2; shift; shift; timeout 600 env "$@" python bench.py --config $cfg $Q $EXTRA > $OUT/${cfg}_$tag.json 2> $OUT/${cfg}_$tag.err; python - <<PY
import json
try:
    d = json.load(open("$OUT/${cfg}_$tag.json")); r = d["roofline"]
    print("%s %-18s %8.1f M/s frac %.3f in_flight %5d launch %6.2f ms  eps %d  kernel/step %.2f  S %.2f U %.2f" % ("$cfg", "$tag", d["value"]/1e6, r["frac"], r["interactions_in_flight"], r["avg_launch_ms"], d["config"]["epochs_per_step"], r["kernel_time_fraction_of_step"], r["draws_per_interaction"], r["updates_per_interaction"]))
except Exception as e:
    print("$cfg $tag FAILED", e)
PY
}
EXTRA="--steps 2 --warmup 1 --epochs-per-step 2" run c3 default A=1
EXTRA="--steps 2 --warmup 1 --epochs-per-step 1 --scale 0.25" run c5shard feat A=1
EXTRA="--steps 2 --warmup 1 --epochs-per-step 1 --scale 0.25 --feat-kernel 1" run c5shard generic A=1
timeout 900 python bench.py > $OUT/bench_c2_full.json 2> $OUT/bench_c2_full.err; tail -4 $OUT/bench_c2_full.err; cat $OUT/bench_c2_full.json
timeout 900 python bench.py --config c3 > $OUT/bench_c3_full.json 2> $OUT/bench_c3_full.err; tail -4 $OUT/bench_c3_full.err; cat $OUT/bench_c3_full.json
timeout 900 python bench.py --config c4shard > $OUT/bench_c4_full.json 2> $OUT/bench_c4_full.err; tail -3 $OUT/bench_c4_full.err; cat $OUT/bench_c4_full.json
