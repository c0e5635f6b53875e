#!/bin/bash
# Instruction mix of the bench kernel (one PMC pass).  Usage: tools/pmc_insts.sh [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_insts
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/a -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $R && python - <<PY
import sqlite3, json
con = sqlite3.connect("$OUT/a/pmc_results.db")
rows = con.execute("select k.name, p.counter_name, count(distinct p.dispatch_id), sum(p.counter_value) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name").fetchall()
b = json.load(open("$OUT/bench.json"))
n = b["config"]["workload"]
per = 20000263 / b["roofline"]["launches_per_epoch"]
for name, c, nd, v in rows:
    if "fit_" in name: print("%-22s %10.1f per interaction" % (c, v / nd / per))
print("%.1f M/s" % (b["value"] / 1e6))
PY
