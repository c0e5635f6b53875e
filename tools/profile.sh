#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun).  Usage: tools/profile.sh <tag> [bench args]
# 1. kernel trace + stats of the bench command
# 2. PMC passes (separate runs, --kernel-trace only): FETCH_SIZE, WRITE_SIZE, SQ mix, TCC
TAG=${1:-r01}
shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="--steps 2 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline $* > $OUT/bench_trace.json 2> $OUT/bench_trace.err
pass() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$n -o pmc -- python $R/bench.py $SHORT > $OUT/bench_$n.json 2> $OUT/bench_$n.err; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA0_RDREQ_sum
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC
find $OUT -name "*.csv" | xargs ls -la 2>/dev/null | head
# keep the merged directory small: drop anything huge
find $OUT -size +20M -delete
cd $R && python tools/prof_summary.py $TAG > $OUT/summary.txt 2>&1
tail -5 $OUT/summary.txt
