#!/bin/bash
# One GPU-box visit: parity tests, then short bench runs of the tile and generic WARP kernels.
# Usage (through gpurun): bash tools/gpu_check.sh [tag]
TAG=${1:-check}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_tile.json 2> $OUT/bench_tile.err
tail -3 $OUT/bench_tile.err; cat $OUT/bench_tile.json
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --warp-kernel 1 > $OUT/bench_generic.json 2> $OUT/bench_generic.err
cat $OUT/bench_generic.json
