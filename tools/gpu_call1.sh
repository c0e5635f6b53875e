#!/bin/bash
# Round 2, GPU visit 1: the whole -m gpu suite (no -x: every failure is wanted), then short bench
# runs of the named configs with the new and the generic kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02a
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=25 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -60 $OUT/pytest.log
B="--steps 3 --warmup 1 --no-cpu-baseline --no-quality --no-fit"
timeout 300 python bench.py $B > $OUT/bench_c2.json 2> $OUT/bench_c2.err; tail -2 $OUT/bench_c2.err; cat $OUT/bench_c2.json
timeout 300 python bench.py $B --config c3 > $OUT/bench_c3_feat.json 2> $OUT/bench_c3_feat.err; tail -2 $OUT/bench_c3_feat.err; cat $OUT/bench_c3_feat.json
timeout 300 python bench.py $B --config c3 --feat-kernel 1 > $OUT/bench_c3_generic.json 2> $OUT/bench_c3_generic.err; tail -2 $OUT/bench_c3_generic.err; cat $OUT/bench_c3_generic.json
timeout 400 python bench.py $B --config c4shard > $OUT/bench_c4.json 2> $OUT/bench_c4.err; tail -2 $OUT/bench_c4.err; cat $OUT/bench_c4.json
