#!/bin/bash
# Round 2, GPU visit 2: full suite again (+ the parity deltas), ranks timing (MFMA vs scalar), C3 sweeps of the
# row-stream kernel, multi-GPU emulation, hybrid precision study.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02b
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -rP -p no:cacheprovider > $OUT/pytest_full.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_full.log
grep -E "delta|passed|failed|^FAILED|^ERROR" $OUT/pytest_full.log > $OUT/pytest.log
tail -40 $OUT/pytest.log
timeout 300 python tools/ranks_timing.py > $OUT/ranks_mfma.txt 2>&1; tail -3 $OUT/ranks_mfma.txt
LIGHTFM_AMD_RANKS_MFMA=0 timeout 300 python tools/ranks_timing.py > $OUT/ranks_scalar.txt 2>&1; tail -3 $OUT/ranks_scalar.txt
B="--steps 2 --warmup 1 --no-cpu-baseline --no-quality --no-fit --config c3 --epochs-per-step 2"
run() { tag=$1; shift; timeout 300 env "$@" python bench.py $B $EXTRA > $OUT/c3_$tag.json 2> $OUT/c3_$tag.err; python - <<PY
import json
try:
    d = json.load(open("$OUT/c3_$tag.json")); r = d["roofline"]
    print("c3 %-22s %7.1f M/s frac %.3f in_flight %d launch %.2f ms" % ("$tag", d["value"]/1e6, r["frac"], r["interactions_in_flight"], r["avg_launch_ms"]))
except Exception as e:
    print("c3 $tag FAILED", e)
PY
}
EXTRA="" run default A=1
EXTRA="" run lds8 LIGHTFM_AMD_FEAT_LDS_KB=8
EXTRA="" run lds16 LIGHTFM_AMD_FEAT_LDS_KB=16
EXTRA="" run lds22 LIGHTFM_AMD_FEAT_LDS_KB=22
EXTRA="" run lds12_wpb2 LIGHTFM_AMD_FEAT_WAVES_PER_BLOCK=2
EXTRA="" run lds12_wpb4 LIGHTFM_AMD_FEAT_WAVES_PER_BLOCK=4
EXTRA="--update-mode 2" run nowrite A=1
EXTRA="--update-mode 1" run stores A=1
EXTRA="--debug 8" run f32math A=1
EXTRA="" run cached LIGHTFM_AMD_TABLE_ALLOC=0
EXTRA="--shared-cap -1" run nocap A=1
EXTRA="--feat-kernel 1" run generic A=1
S="1,2,3,4,5,6,7,8,9,10"
QUALITY_TAGS=40,4 QUALITY_SEEDS=$S QUALITY_MODES=3 timeout 600 python tools/quality.py ml-100k 10 32 warp 0 > $OUT/hybrid100k_auto.txt 2>&1; tail -3 $OUT/hybrid100k_auto.txt
QUALITY_TAGS=40,4 QUALITY_SEEDS=$S QUALITY_MODES=3 QUALITY_REF_THREADS=16 LIGHTFM_AMD_SHARED_CAP=-1 timeout 600 python tools/quality.py ml-100k 10 32 warp 0 > $OUT/hybrid100k_nocap.txt 2>&1; tail -1 $OUT/hybrid100k_nocap.txt
QUALITY_TAGS=40,4 QUALITY_SEEDS=$S QUALITY_MODES=3 QUALITY_REF_THREADS=16 LIGHTFM_AMD_SHARED_CAP=64 timeout 600 python tools/quality.py ml-100k 10 32 warp 0 > $OUT/hybrid100k_cap64.txt 2>&1; tail -1 $OUT/hybrid100k_cap64.txt
EMU_SEEDS=1,2 EMU_EPOCHS=5 timeout 1200 python tools/multi_gpu_emulation.py 1:sum:4:16384:0 8:sum:4:16384:0 8:mean:4:16384:0 8:adagrad:4:16384:0 8:adagrad:4:16384:2097152 8:adagrad:4:16384:20000000 8:sum:4:16384:2097152 8:adagrad:1000000000:20000000:20000000 > $OUT/emulation.txt 2>&1
grep "^K=" $OUT/emulation.txt; tail -3 $OUT/emulation.txt
