#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun).  Usage: tools/profile.sh <tag>
# 1. kernel trace + stats of the default bench command
# 2. PMC passes (separate runs, --kernel-trace only): FETCH_SIZE, WRITE_SIZE, SQ mix
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py > $OUT/bench_trace.json 2> $OUT/bench_trace.err
SHORT="--steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python $R/bench.py $SHORT > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python $R/bench.py $SHORT > $OUT/bench_write.json 2> $OUT/bench_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/pmc_sq -o pmc -- python $R/bench.py $SHORT > $OUT/bench_sq.json 2> $OUT/bench_sq.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d $OUT/pmc_sq2 -o pmc -- python $R/bench.py $SHORT > $OUT/bench_sq2.json 2> $OUT/bench_sq2.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum -d $OUT/pmc_tcc -o pmc -- python $R/bench.py $SHORT > $OUT/bench_tcc.json 2> $OUT/bench_tcc.err
rocprofv3 -L > $OUT/counters_list.txt 2>&1
find $OUT -name "*.csv" | xargs ls -la
# keep the merged directory small: drop anything huge
find $OUT -size +20M -delete
