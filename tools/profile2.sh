#!/bin/bash
# Lean profiling recipe (GPU box): kernel trace + stats, then three PMC passes (separate runs, --kernel-trace
# only: FETCH_SIZE; WRITE_SIZE; TCC hit / miss / atomic / EA read requests).  The rocpd databases are reduced to
# the small summaries of tools/prof_summary.py (copied to gpurun_out/<tag>_*), then deleted.
# Usage: tools/profile2.sh <tag> <bench args>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (round 5: a step is one epoch of a fresh fit; the driver's K / W, one fit, no reporting legs)
B="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 0 --steps ${PROF_STEPS:-20} --warmup ${PROF_WARMUP:-5} $*"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py $B > $OUT/bench_trace.json 2> $OUT/bench_trace.err
pass() { n=$1; shift; timeout -k 5 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$n -o pmc -- python $R/bench.py $B > $OUT/bench_$n.json 2> $OUT/bench_$n.err; }
if [ -z "$TRACE_ONLY" ]; then
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA0_RDREQ_sum
# request sizes at the memory side: calibrates FETCH_SIZE (which tallies every read request at 64 B) for kernels whose
# requests are not all 128-byte ones (tools/prof_summary.py: hbm_bytes_per_launch_calibrated)
pass rdsz TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum
pass wrsz TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum
fi
cd $R && python tools/prof_summary.py $TAG > $OUT/summary.txt 2>&1
cp $R/profiles/${TAG}_* $R/gpurun_out/ 2>/dev/null
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -delete; find $OUT -size +2M -delete
tail -3 $OUT/summary.txt | cut -c1-300
