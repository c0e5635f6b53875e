#!/bin/bash
# Round 2, GPU visit 3: suite, C2 / C3 / C5 experiments, full bench lines, kernel-trace profiles.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02c
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -rP -p no:cacheprovider > $OUT/pytest_full.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_full.log
grep -E "delta|passed|failed|^FAILED|^ERROR" $OUT/pytest_full.log > $OUT/pytest.log
tail -15 $OUT/pytest.log
Q="--no-cpu-baseline --no-quality --no-fit"
run() { cfg=$1; tag=$2; shift; shift; timeout 400 env "$@" python bench.py --config $cfg $Q $EXTRA > $OUT/${cfg}_$tag.json 2> $OUT/${cfg}_$tag.err; python - <<PY
import json
try:
    d = json.load(open("$OUT/${cfg}_$tag.json")); r = d["roofline"]
    print("%s %-18s %8.1f M/s frac %.3f in_flight %5d launch %6.2f ms  eps %d  kernel/step %.2f  S %.2f U %.2f" % ("$cfg", "$tag", d["value"]/1e6, r["frac"], r["interactions_in_flight"], r["avg_launch_ms"], d["config"]["epochs_per_step"], r["kernel_time_fraction_of_step"], r["draws_per_interaction"], r["updates_per_interaction"]))
except Exception as e:
    print("$cfg $tag FAILED", e)
PY
}
EXTRA="--steps 3 --warmup 1 --epochs-per-step 8" run c2 default A=1
EXTRA="--steps 3 --warmup 1 --epochs-per-step 8 --debug 16" run c2 scalarbias A=1
EXTRA="--steps 3 --warmup 1 --epochs-per-step 8 --update-mode 2" run c2 nowrite A=1
EXTRA="--steps 3 --warmup 1 --epochs-per-step 8 --emulate-shard 8" run c2 shard8_auto A=1
EXTRA="--steps 3 --warmup 1 --epochs-per-step 8 --emulate-shard 8" run c2 shard8_uncached LIGHTFM_AMD_TABLE_ALLOC=3
EXTRA="--steps 3 --warmup 1 --epochs-per-step 8 --update-mode 2" run c4shard nowrite A=1
for mw in 1024 1536 2048 2560; do
EXTRA="--steps 2 --warmup 1 --epochs-per-step 2 --max-waves $mw" run c3 mw$mw A=1
done
EXTRA="--steps 2 --warmup 1 --epochs-per-step 2 --max-waves 1792" run c3 mw1792_lds22 LIGHTFM_AMD_FEAT_LDS_KB=22
EXTRA="--steps 2 --warmup 1 --epochs-per-step 1 --scale 0.1" run c5shard feat A=1
EXTRA="--steps 2 --warmup 1 --epochs-per-step 1 --scale 0.1 --feat-kernel 1" run c5shard generic A=1
EXTRA="--steps 2 --warmup 1 --epochs-per-step 1 --scale 0.1 --update-mode 2" run c5shard nowrite A=1
# full bench lines (quality + CPU baseline + end-to-end fit)
timeout 900 python bench.py > $OUT/bench_c2_full.json 2> $OUT/bench_c2_full.err; tail -4 $OUT/bench_c2_full.err; cat $OUT/bench_c2_full.json
timeout 900 python bench.py --config c3 > $OUT/bench_c3_full.json 2> $OUT/bench_c3_full.err; tail -4 $OUT/bench_c3_full.err; cat $OUT/bench_c3_full.json
timeout 900 python bench.py --config c4shard > $OUT/bench_c4_full.json 2> $OUT/bench_c4_full.err; tail -3 $OUT/bench_c4_full.err; cat $OUT/bench_c4_full.json
# kernel-trace profiles of the same commands (short)
cd /tmp && export TMPDIR=/tmp
for cfg in c2 c3 c4shard; do
rocprofv3 --kernel-trace --stats -d $OUT/trace_$cfg -o trace -- python $R/bench.py --config $cfg $Q --steps 3 --warmup 1 --epochs-per-step 2 > $OUT/trace_$cfg.json 2> $OUT/trace_$cfg.err
done
find $OUT -name "*_kernel_stats.csv" | head; find $OUT -size +8M -delete
for cfg in c2 c3 c4shard; do f=$(find $OUT/trace_$cfg -name "*kernel_stats.csv" | head -1); echo "== $cfg $f"; head -8 "$f"; done
