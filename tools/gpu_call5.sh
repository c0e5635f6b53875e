#!/bin/bash
# Round 2, GPU visit 5 (first of the re-entered session): memory-system microbenchmarks, the full GPU
# suite, bench lines + ablations of the four configurations (incl. the per-phase profiling builds),
# the multi-GPU emulation, the full c2 bench line, predict_ranks timing.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02e
mkdir -p $OUT
cd $R
timeout 120 tools/_bin/membench > $OUT/membench.txt 2>&1; tail -5 $OUT/membench.txt
timeout 1300 python -m pytest tests -m gpu -q -rP --durations=12 -p no:cacheprovider > $OUT/pytest_full.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_full.log
grep -E "delta|passed|failed|^FAILED|^ERROR|s call|s setup|pytest exit" $OUT/pytest_full.log > $OUT/pytest.log
tail -30 $OUT/pytest.log
Q="--no-cpu-baseline --no-quality --no-fit"
run() { cfg=$1; tag=$2; shift; shift; timeout 400 env "$@" python bench.py --config $cfg $Q $EXTRA > $OUT/${cfg}_$tag.json 2> $OUT/${cfg}_$tag.err; python - <<PY
import json
try:
    d = json.load(open("$OUT/${cfg}_$tag.json")); r = d["roofline"]
    print("%s %-14s %8.1f M/s frac %.3f in_flight %5d launch %6.2f ms eps %d k/step %.2f S %.2f U %.2f %s" % ("$cfg", "$tag", d["value"]/1e6, r["frac"], r["interactions_in_flight"], r["avg_launch_ms"], d["config"]["epochs_per_step"], r["kernel_time_fraction_of_step"], r["draws_per_interaction"], r["updates_per_interaction"], r.get("phase_cycles_per_pass") or r.get("phase_cycles_per_interaction") or ""))
except Exception as e:
    print("$cfg $tag FAILED", e)
PY
}
E8="--steps 3 --warmup 1 --epochs-per-step 8"
EXTRA="$E8" run c2 default A=1
EXTRA="$E8 --warp-kernel 2" run c2 timed A=1
EXTRA="$E8 --warp-kernel 2 --debug 1" run c2 timed_drain A=1
EXTRA="$E8 --update-mode 2" run c2 nowrite A=1
EXTRA="$E8 --first-batch 5" run c2 fb5 A=1
EXTRA="$E8 --first-batch 3" run c2 fb3 A=1
E2="--steps 2 --warmup 1 --epochs-per-step 2"
EXTRA="$E2" run c3 default A=1
EXTRA="$E2 --feat-kernel 2" run c3 timed A=1
EXTRA="$E2 --feat-kernel 2 --debug 1" run c3 timed_drain A=1
EXTRA="$E2 --update-mode 2" run c3 nowrite A=1
EXTRA="$E2 --debug 8" run c3 f32math A=1
EXTRA="$E2 --max-waves 2048" run c3 mw2048 A=1
EXTRA="$E2 --feat-kernel 1" run c3 generic A=1
EXTRA="$E2" run c3 cached LIGHTFM_AMD_TABLE_ALLOC=0
EXTRA="$E2" run c4shard default A=1
E1="--steps 2 --warmup 1 --epochs-per-step 1 --scale 0.25"
EXTRA="$E1" run c5shard default A=1
EXTRA="$E1 --feat-kernel 2" run c5shard timed A=1
EXTRA="$E1 --update-mode 2" run c5shard nowrite A=1
EXTRA="$E1 --feat-kernel 1" run c5shard generic A=1
EMU_SEEDS=1,2 EMU_EPOCHS=5 timeout 900 python tools/multi_gpu_emulation.py 1:sum:4:16384:0 8:adagrad:4:16384:0 8:sum:4:16384:0 2:adagrad:4:16384:0 > $OUT/emulation.txt 2>&1
grep "^K=" $OUT/emulation.txt; tail -2 $OUT/emulation.txt
timeout 900 python bench.py > $OUT/bench_c2_full.json 2> $OUT/bench_c2_full.err; tail -4 $OUT/bench_c2_full.err; cat $OUT/bench_c2_full.json
timeout 300 python tools/ranks_timing.py > $OUT/ranks_mfma.txt 2>&1; tail -2 $OUT/ranks_mfma.txt
