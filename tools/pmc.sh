#!/bin/bash
# Extra PMC passes for the memory-side analysis (GPU box).  Usage: tools/pmc.sh <tag> [bench args]
TAG=${1:-mem}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="--steps 2 --warmup 1 --no-cpu-baseline $*"
pass() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$n -o pmc -- python $R/bench.py $SHORT > $OUT/bench_$n.json 2> $OUT/bench_$n.err; }
pass a TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum
pass b TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum TCC_EA0_WRREQ_sum
pass c TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum
pass d TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
pass e TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum
pass f TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_LATENCY_FIFO_FULL_sum
pass g TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum
cd $R && python - <<PY
import sqlite3, os, json
out = {}
base = "$OUT"
for sub in sorted(os.listdir(base)):
    db = os.path.join(base, sub, "pmc_results.db")
    if not os.path.exists(db): continue
    con = sqlite3.connect(db)
    rows = con.execute("select k.name, p.counter_name, count(distinct p.dispatch_id), sum(p.counter_value) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name").fetchall()
    for name, c, n, v in rows:
        if "fit_" in name: out[c] = v / n
json.dump(out, open(os.path.join(base, "summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
find $OUT -name "*.db" -size +5M -delete
