#!/bin/bash
# Ablation matrix of the WARP kernel on the bench workload (GPU box).  Each argument is one
# quoted set of bench.py flags.
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $*"; python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('   %.1f M/s  %.2f ms/epoch  launch %.3f ms  frac %.3f  draws %.2f upd %.2f' % (d['value']/1e6, d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['draws_per_interaction'], r['updates_per_interaction']))
if 'phase_cycles_per_pass' in r: print('  ', r['phase_cycles_per_pass'])"; }
for args in "$@"; do run $args; done
