#!/bin/bash
# tools/visit.sh <step> [args] -- everything this repo runs on the GPU box, one parametrised script
# (gpurun --timeout N -- 'bash tools/visit.sh <step>').  Outputs land in gpurun_out/<step>/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
STEP=${1:-suite}; shift
OUT=$R/gpurun_out/$STEP
mkdir -p $OUT
export PYTHONFAULTHANDLER=1
PYT="python3 -m pytest -p no:cacheprovider"
summ() { grep -aE "abort_trace|Memory access fault|callbackQueue|HSA_STATUS|hipError|VALIDATE|FAILED|ERROR|passed|failed|Fatal|Aborted|SIGSEGV|SIGABRT|received signal|malloc|free\(\)|corrupt" "$1" | cut -c1-400 | head -${2:-30}; }

case $STEP in
hunt)
  # The process abort.  -s: no fd capture, so ROCr / ROCclr / glibc messages reach the log.
  export LD_PRELOAD=$R/tools/_bin/libaborttrace.so ABORT_TRACE_FILE=$OUT/abort_trace
  died=0
  for i in 1 2 3; do
    timeout -k 5 400 $PYT tests -m gpu -x -q -s --deselect tests/test_precision_parity.py > $OUT/plain_$i.log 2>&1
    rc=$?; echo "plain run $i: exit $rc  $(grep -aE ' passed| failed' $OUT/plain_$i.log | tail -1)"
    if [ $rc -ne 0 ]; then died=1; summ $OUT/plain_$i.log; tail -c 1500 $OUT/plain_$i.log; break; fi
  done
  dmesg 2>/dev/null | tail -30 > $OUT/dmesg.txt
  unset LD_PRELOAD
  # the same under rocgdb: a GPU memory fault stops at the faulting wave (kernel name + pc), a host abort at its frame
  for i in 1 2; do
    timeout -k 5 600 rocgdb -batch -ex "set pagination off" -ex "set confirm off" -ex "handle SIGUSR1 nostop noprint pass" \
      -ex run -ex "echo \n==== STOPPED ====\n" -ex "info threads" -ex "bt 40" -ex "x/16i \$pc-32" -ex "info agents" -ex "info dispatches" \
      -ex "thread apply all bt 12" \
      --args python3 -m pytest -p no:cacheprovider tests -m gpu -x -q -s --deselect tests/test_precision_parity.py > $OUT/rocgdb_$i.log 2>&1
    echo "rocgdb run $i: exit $?  $(grep -aE ' passed| failed' $OUT/rocgdb_$i.log | tail -1)"
    if grep -aq "==== STOPPED" $OUT/rocgdb_$i.log && grep -aqE "received signal|Memory access" $OUT/rocgdb_$i.log; then
      sed -n '/received signal/,$p' $OUT/rocgdb_$i.log | cut -c1-300 | head -120; break
    fi
  done
  ls -la $OUT | head -30
  for f in $OUT/abort_trace.*; do [ -s "$f" ] && { echo "--- $f"; head -60 "$f"; }; done
  ;;
hunt2)
  # precise fault location: debug-symbol build (LFM_BUILD_DEBUG=1), allocation / launch trace, rocgdb precise memory
  export LIGHTFM_AMD_TRACE=1
  for i in 1 2 3; do
    timeout -k 5 900 rocgdb -batch -ex "set pagination off" -ex "set confirm off" -ex "set amdgpu precise-memory on" \
      -ex "handle SIGUSR1 nostop noprint pass" -ex run -ex "echo \n==== STOPPED ====\n" -ex "bt 12" -ex "x/10i \$pc-32" \
      -ex "info dispatches" -ex "p a" -ex "p/x a.m" -ex "p \$exec" \
      -ex "p/x \$s0" -ex "p/x \$s1" -ex "p/x \$s8" -ex "p/x \$s9" -ex "p/x \$s10" -ex "p/x \$s11" -ex "p/x \$s22" -ex "p/x \$s34" -ex "p/x \$s35" \
      -ex "p/x \$s72" -ex "p/x \$s73" -ex "p/x \$s74" -ex "p/x \$s75" \
      -ex "p/x \$v10" -ex "p/x \$v11" -ex "p/x \$v18" -ex "p/x \$v19" -ex "p/x \$v20" -ex "p/x \$v21" -ex "p/x \$v76" -ex "p/x \$v77" -ex "p/x \$v78" -ex "p/x \$v79" -ex "p/x \$v96" -ex "p/x \$v164" \
      -ex "info registers" \
      --args python3 -m pytest -p no:cacheprovider tests -m gpu -x -q -s --deselect tests/test_precision_parity.py > $OUT/rocgdb_full_$i.log 2>&1
    echo "rocgdb precise run $i: exit $?  $(grep -aE ' passed| failed' $OUT/rocgdb_full_$i.log | tail -1)"
    grep -av "^LFM_LAUNCH\|^\[New Thread\|^\[Thread" $OUT/rocgdb_full_$i.log | head -c 3000000 > $OUT/rocgdb_$i.log
    grep -a "^LFM_LAUNCH" $OUT/rocgdb_full_$i.log | tail -400 > $OUT/launches_$i.log
    grep -a "^LFM_ALLOC\|^LFM_FREE" $OUT/rocgdb_full_$i.log | tail -3000 > $OUT/allocs_$i.log
    rm -f $OUT/rocgdb_full_$i.log
    if grep -aq "received signal" $OUT/rocgdb_$i.log; then sed -n '/received signal/,$p' $OUT/rocgdb_$i.log | cut -c1-400 | head -150; break; fi
  done
  unset LIGHTFM_AMD_TRACE
  # discriminators: does the fault survive (a) serialised kernels, (b) blit copies instead of SDMA, (c) a sync after every launch
  for cfg in "AMD_SERIALIZE_KERNEL=3" "HSA_ENABLE_SDMA=0" "LIGHTFM_AMD_REG_SYNC=1" "PLAIN=1"; do for i in 1 2; do
    env $cfg timeout -k 5 400 $PYT tests -m gpu -x -q -s --deselect tests/test_precision_parity.py > $OUT/cfg_${cfg%%=*}_$i.log 2>&1
    echo "$cfg run $i: exit $?  $(grep -aE ' passed| failed|Memory access fault' $OUT/cfg_${cfg%%=*}_$i.log | tail -2 | tr '\n' ' ')"
  done; done
  ;;
rootcause)
  # the pool OFF (hipMalloc / hipFree per buffer, the round-2 behaviour): with uncached tables (default) and without
  for cfg in "LIGHTFM_AMD_POOL=0" "LIGHTFM_AMD_POOL=0 LIGHTFM_AMD_TABLE_ALLOC=0"; do for i in 1 2 3; do
    tag=$(echo "$cfg" | tr ' =' '__')
    env $cfg timeout -k 5 400 $PYT tests -m gpu -x -q -s --deselect tests/test_precision_parity.py > $OUT/${tag}_$i.log 2>&1
    echo "$cfg run $i: exit $?  $(grep -aE ' passed| failed|Memory access fault' $OUT/${tag}_$i.log | tail -2 | tr '\n' ' ')"
    grep -aE "^FAILED|^ERROR|LFM_ECORRUPT|shuffle entry" $OUT/${tag}_$i.log | cut -c1-300 | head -4
  done; done
  ;;
membench)
  tools/_bin/membench "$@" > $OUT/membench.txt 2>&1; echo "membench exit $?"; grep -a "atomic " $OUT/membench.txt | tail -40
  ;;
suite)
  # the driver's round-end command, N times (default 3), uncaptured
  N=${1:-3}
  for i in $(seq 1 $N); do
    timeout -k 5 1500 $PYT tests -m gpu -x -q -s > $OUT/suite_$i.log 2>&1
    echo "suite run $i: exit $?  $(grep -aE ' passed| failed' $OUT/suite_$i.log | tail -1)"; summ $OUT/suite_$i.log 12
  done
  ;;
r3a)
  # first visit of round 3 after the re-entry: suite x3, atomic flavours, C2 line (alpha = 0 and 1e-6), C3 short
  for i in 1 2 3; do
    timeout -k 5 900 $PYT tests -m gpu -x -q -s > $OUT/suite_$i.log 2>&1
    echo "suite run $i: exit $?  $(grep -aE ' passed| failed' $OUT/suite_$i.log | tail -1)"; summ $OUT/suite_$i.log 12
  done
  timeout 200 tools/_bin/membench > $OUT/membench.txt 2>&1; echo "membench exit $?"; grep -a "waves/CU [fu]" $OUT/membench.txt | head -40
  timeout 400 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench c2 exit $?"; cut -c1-1500 $OUT/bench_c2.json
  S="--no-cpu-baseline --no-quality --no-fit --steps 6 --warmup 2"
  timeout 200 python bench.py $S --item-alpha 1e-6 --user-alpha 1e-6 > $OUT/bench_c2_reg.json 2> $OUT/bench_c2_reg.err; echo "bench c2 reg exit $?"; cut -c1-700 $OUT/bench_c2_reg.json
  timeout 300 python bench.py $S --config c3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "bench c3 exit $?"; cut -c1-700 $OUT/bench_c3.json
  timeout 300 python bench.py $S --config c3 --item-alpha 1e-6 --user-alpha 1e-6 > $OUT/bench_c3_reg.json 2> $OUT/bench_c3_reg.err; echo "bench c3 reg exit $?"; cut -c1-700 $OUT/bench_c3_reg.json
  ;;
r3b)
  # RegScale v3 (extrapolating readers, publish at wave end), scoring session, hoisted-pointer fix
  N=${1:-1}
  for i in $(seq 1 $N); do
    timeout -k 5 900 $PYT tests -m gpu -x -q -s > $OUT/suite_$i.log 2>&1
    echo "suite run $i: exit $?  $(grep -aE ' passed| failed' $OUT/suite_$i.log | tail -1)"; summ $OUT/suite_$i.log 12
  done
  S="--no-cpu-baseline --no-quality --no-fit --steps 6 --warmup 2"
  for cfg in "c2" "c2 --item-alpha 1e-6 --user-alpha 1e-6" "c3" "c3 --item-alpha 1e-6 --user-alpha 1e-6" "c4shard" "c4shard --item-alpha 1e-6 --user-alpha 1e-6"; do
    tag=$(echo "$cfg" | tr ' ' '_' | tr -d '-')
    timeout 300 python bench.py $S --config $cfg > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; echo "bench $cfg exit $?"
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$tag.json"))
    r = d["roofline"]
    print("  %.1f M/s  frac %.3f  atomic %.3f  launch %.3f ms  U %.3f  kernel %s" % (d["value"] / 1e6, r["frac"], r["atomic_unit"]["frac"], r["avg_launch_ms"], r["updates_per_interaction"], r["kernel"]))
except Exception as e:
    print("  no result:", e)
PY
  done
  ;;
final)
  # the driver's sequence: suite (N times, default 2), smoke, default bench; then rocprofv3 trace + PMC of the
  # configs named after N (default c2 c3 c4shard):   tools/visit.sh final [N [cfg ...]]
  NS=${1:-2}; shift; CFGS=${*:-c2 c3 c4shard}
  for i in $(seq 1 $NS); do
    timeout -k 5 1200 $PYT tests -m gpu -x -q > $OUT/suite_$i.log 2>&1
    echo "suite run $i: exit $?  $(grep -aE ' passed| failed' $OUT/suite_$i.log | tail -1)"; summ $OUT/suite_$i.log 8
  done
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  ( time timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_default.json"))
    def show(n, v, r):
        print("  %-8s %8.1f M/s  frac %.3f  atomic %.3f  launch %.3f ms  U %.3f  traffic %s  %s" % (n, v / 1e6, r["frac"], r["atomic_unit"]["frac"], r["avg_launch_ms"], r["updates_per_interaction"], r.get("traffic_over_algorithmic"), r["kernel"]))
    show("c2", d["value"], d["roofline"])
    for e in d.get("extra_configs", []):
        if "error" in e: print("  ", e)
        else: show(e["name"], e["value"], e["roofline"])
    q = d.get("quality") or {}
    print("  quality", q.get("precision_at_10"), q.get("precision_at_10_ref"), q.get("delta"))
    print("  cpu", (d.get("cpu_baseline") or {}).get("value"), "fit", (d.get("end_to_end_fit") or {}).get("value"))
except Exception as e:
    print("  no result:", e)
PY
  for cfg in $CFGS; do bash tools/profile2.sh r03_$cfg --config $cfg; done
  cp $R/profiles/r03_c*_kernel_stats.txt $R/profiles/r03_c*_pmc_summary.json $OUT/ 2>/dev/null
  ;;
r3c)
  # suite, the driver's default bench line (with the extra_configs legs), rocprofv3 trace + PMC of c2 / c3 / c4shard
  timeout -k 5 1200 $PYT tests -m gpu -x -q -s > $OUT/suite_1.log 2>&1
  echo "suite: exit $?  $(grep -aE ' passed| failed' $OUT/suite_1.log | tail -1)"; summ $OUT/suite_1.log 12
  grep -a "per side, fixed" $OUT/suite_1.log | cut -c1-200
  ( time timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real; echo "bench default exit $?"
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_default.json"))
    def show(n, v, r):
        print("  %-8s %8.1f M/s  frac %.3f  atomic %.3f  launch %.3f ms  U %.3f  %s" % (n, v / 1e6, r["frac"], r["atomic_unit"]["frac"], r["avg_launch_ms"], r["updates_per_interaction"], r["kernel"]))
    show("c2", d["value"], d["roofline"])
    for e in d.get("extra_configs", []):
        if "error" in e: print("  ", e)
        else: show(e["name"], e["value"], e["roofline"])
    print("  quality", d.get("quality"))
    print("  cpu", (d.get("cpu_baseline") or {}).get("value"), "fit", (d.get("end_to_end_fit") or {}).get("value"))
except Exception as e:
    print("  no result:", e)
PY
  tail -5 $OUT/bench_default.err | cut -c1-300
  for cfg in c2 c3 c4shard; do bash tools/profile2.sh r03_$cfg --config $cfg; done
  cp $R/profiles/r03_* $OUT/ 2>/dev/null
  ;;
r4a)
  # round 4, first visit: the new exact-parity tests at the BASELINE shapes, the reference's own suite through the shim,
  # C3 A/B of the round-2 tree (tools/_bin/r2tree, built from commit b0bf975) against the current one at the same
  # epochs, C5 shard evidence (kernel trace + FETCH/WRITE counters at --scale 0.1; per-phase split at full size)
  timeout -k 5 900 $PYT tests/test_baseline_shapes.py tests/test_reference_suite.py -m gpu -q -s > $OUT/new_tests.log 2>&1
  echo "new tests: exit $?  $(grep -aE ' passed| failed' $OUT/new_tests.log | tail -1)"; summ $OUT/new_tests.log 20
  grep -aE "^E  |differ|assert" $OUT/new_tests.log | cut -c1-300 | head -40
  S="--no-cpu-baseline --no-quality --no-fit --steps 3 --warmup 1 --epochs-per-step 2 --config c3"
  for i in 1 2; do
    ( cd $R/tools/_bin/r2tree && timeout 300 python bench.py $S > $OUT/c3_r2_$i.json 2> $OUT/c3_r2_$i.err )
    timeout 300 python bench.py $S > $OUT/c3_r3_$i.json 2> $OUT/c3_r3_$i.err
    for t in r2 r3; do python - <<PY
import json
try:
    d = json.load(open("$OUT/c3_${t}_$i.json")); r = d["roofline"]
    print("  c3 %s run $i: %8.2f M/s  frac %.3f  atomic %.3f  launch %.3f ms  in_flight %s  %s" % ("$t", d["value"] / 1e6, r["frac"], r["atomic_unit"]["frac"], r["avg_launch_ms"], r.get("interactions_in_flight"), d["config"].get("timed_epochs")))
except Exception as e:
    print("  c3 $t run $i: no result:", e)
PY
    done
  done
  TRACE_ONLY= bash tools/profile2.sh r04_c5shard --config c5shard --scale 0.1
  timeout 400 python bench.py --config c5shard --feat-kernel 2 --no-cpu-baseline --no-quality --no-fit --steps 2 --warmup 1 --epochs-per-step 1 > $OUT/c5_phases.json 2> $OUT/c5_phases.err
  echo "c5 phases exit $?"; python - <<PY
import json
try:
    d = json.load(open("$OUT/c5_phases.json")); r = d["roofline"]
    print("  c5shard timed build: %.2f M/s frac %.3f" % (d["value"] / 1e6, r["frac"]), r.get("phase_cycles_per_interaction"))
except Exception as e:
    print("  no result:", e)
PY
  cp $R/profiles/r04_* $OUT/ 2>/dev/null
  ;;
r4b)
  # round 4, second visit: Bloom pre-filter of in_positives in the tile kernel (default = probed with the candidate rows,
  # --debug 512 = after the scoring pass, --debug 256 = off) and the batched representation reduce of the row-stream
  # kernels, each against the previous commit's library (lightfm_amd/_lib_prev); exactness tests first
  timeout -k 5 900 $PYT tests/test_hip_warp_tile.py tests/test_hip_feat.py tests/test_hip_parity.py tests/test_baseline_shapes.py tests/test_hip_round2.py -m gpu -q -x > $OUT/tests.log 2>&1
  echo "kernel tests: exit $?  $(grep -aE ' passed| failed' $OUT/tests.log | tail -1)"; summ $OUT/tests.log 12
  line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]
    print("  %-34s %9.2f M/s  frac %.3f  atomic %.3f  launch %.3f ms  S %.2f U %.3f  in_flight %s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["atomic_unit"]["frac"], r["avg_launch_ms"], r["draws_per_interaction"], r["updates_per_interaction"], r.get("interactions_in_flight")))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S="--no-cpu-baseline --no-quality --no-fit"
  run() { tag=$1; shift; env $ENVV timeout 400 python bench.py $S "$@" > $OUT/$tag.json 2> $OUT/$tag.err; line $tag $OUT/$tag.json; }
  PREV="LIGHTFM_AMD_LIB=$R/lightfm_amd/_lib_prev/liblfm_hip.so"
  C2="--config c2 --steps 5 --warmup 2 --epochs-per-step 16"
  for i in 1 2; do
    ENVV= run c2_bloom_early_$i $C2
    ENVV= run c2_bloom_late_$i $C2 --debug 512
    ENVV= run c2_bloom_off_$i $C2 --debug 256
    ENVV=$PREV run c2_prev_$i $C2
  done
  C4="--config c4shard --steps 3 --warmup 1 --epochs-per-step 8"
  ENVV= run c4_bloom_early $C4
  ENVV= run c4_bloom_late $C4 --debug 512
  ENVV= run c4_bloom_off $C4 --debug 256
  ENVV=$PREV run c4_prev $C4
  C5="--config c5shard --scale 0.25 --steps 2 --warmup 1 --epochs-per-step 1"
  ENVV= run c5_new $C5
  ENVV=$PREV run c5_prev $C5
  ENVV="LIGHTFM_AMD_FEAT_WAVES_PER_CU=12 LIGHTFM_AMD_FEAT_LDS_KB=13" run c5_new_w12 $C5
  ENVV="LIGHTFM_AMD_FEAT_WAVES_PER_CU=10 LIGHTFM_AMD_FEAT_LDS_KB=16" run c5_new_w10 $C5
  C3="--config c3 --steps 3 --warmup 1 --epochs-per-step 2"
  ENVV= run c3_new $C3
  ENVV=$PREV run c3_prev $C3
  ;;
r4c)
  # round 4, third visit: the gather-ahead tile kernel (default) against the plain one (--debug 1024): exactness and
  # precision@10 gates first, then A/B on c2 / c4shard; C4 two-stage candidate fetch; C5 instruction mix (one PMC pass)
  timeout -k 5 1200 $PYT tests/test_hip_warp_tile.py tests/test_hip_parity.py tests/test_baseline_shapes.py tests/test_hip_round2.py tests/test_precision_parity.py tests/test_lightfm_api.py -m gpu -q -x > $OUT/tests.log 2>&1
  echo "kernel tests: exit $?  $(grep -aE ' passed| failed' $OUT/tests.log | tail -1)"; summ $OUT/tests.log 12
  grep -a "per side, fixed\|precision" $OUT/tests.log | cut -c1-200 | head
  line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]
    print("  %-34s %9.2f M/s  frac %.3f  atomic %.3f  launch %.3f ms  S %.2f U %.3f  in_flight %s  early %.1f M/s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["atomic_unit"]["frac"], r["avg_launch_ms"], r["draws_per_interaction"], r["updates_per_interaction"], r.get("interactions_in_flight"), (d.get("early_epochs") or {}).get("value", 0) / 1e6))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S="--no-cpu-baseline --no-quality --no-fit"
  run() { tag=$1; shift; env $ENVV timeout 400 python bench.py $S "$@" > $OUT/$tag.json 2> $OUT/$tag.err; line $tag $OUT/$tag.json; }
  C2="--config c2 --steps 5 --warmup 2 --epochs-per-step 16"
  for i in 1 2; do
    ENVV= run c2_ahead_$i $C2
    ENVV= run c2_plain_$i $C2 --debug 1024
  done
  C4="--config c4shard --steps 3 --warmup 1 --epochs-per-step 8"
  ENVV= run c4_ahead $C4
  ENVV= run c4_plain $C4 --debug 1024
  ENVV= run c4_plain_fb5 $C4 --debug 1024 --first-batch 5
  ENVV= run c4_ahead_2 $C4
  # quality of the pipelined kernel at the C2 shape (3 seeds, 3 epochs; the reference's number is in every default line)
  timeout 600 python bench.py --config c2 --steps 2 --warmup 1 --epochs-per-step 4 --no-fit > $OUT/c2_quality.json 2> $OUT/c2_quality.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/c2_quality.json")); q = d.get("quality") or {}
    print("  c2 quality (ahead kernel): p@10 %s seeds %s ref %s delta %s; cpu %s" % (q.get("precision_at_10"), q.get("precision_at_10_seeds"), q.get("precision_at_10_ref"), q.get("delta"), (d.get("cpu_baseline") or {}).get("thread_scaling")))
except Exception as e:
    print("  no quality result:", e)
PY
  # C5: instruction mix of the row-stream kernel (one PMC pass, --scale 0.1)
  cd /tmp && export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/c5pmc -o pmc -- python $R/bench.py $S --config c5shard --scale 0.1 --steps 2 --warmup 1 --epochs-per-step 1 > $OUT/c5pmc.json 2> $OUT/c5pmc.err
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SENDMSG -d $OUT/c5pmc2 -o pmc -- python $R/bench.py $S --config c5shard --scale 0.1 --steps 2 --warmup 1 --epochs-per-step 1 > $OUT/c5pmc2.json 2> $OUT/c5pmc2.err
  cd $R && python - <<PY
import sqlite3, json, glob
for sub in ("c5pmc", "c5pmc2"):
    try:
        db = glob.glob("$OUT/%s/**/*results.db" % sub, recursive=True)[0]
        con = sqlite3.connect(db)
        rows = con.execute("select k.name, p.counter_name, count(distinct p.dispatch_id), sum(p.counter_value) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name").fetchall()
        b = json.load(open("$OUT/%s.json" % sub))
        tot = {}
        for name, c, nd, v in rows:
            if "fit_feat" in name: tot[c] = tot.get(c, 0) + v
        n_int = 26001838 * 5.0   # interactions of all profiled epochs (ramp + early + warm-up + timed = 5 epochs at --scale 0.1)
        print("  %s (%.1f M/s):" % (sub, b["value"] / 1e6), {c: round(v / n_int, 1) for c, v in tot.items()}, "per interaction (5 epochs assumed)")
    except Exception as e:
        print("  %s: %r" % (sub, e))
PY
  find $OUT -name "*.db" -size +5M -delete
  ;;
r4d)
  # round 4: the driver's sequence on the current tree -- full GPU suite, smoke, the default bench line (with the
  # extra_configs legs, their cpu baselines, early_epochs, thread scaling)
  timeout -k 5 1500 $PYT tests -m gpu -x -q > $OUT/suite.log 2>&1
  echo "suite: exit $?  $(grep -aE ' passed| failed' $OUT/suite.log | tail -1)"; summ $OUT/suite.log 12
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  ( time timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_default.json"))
    def show(n, v, r, e, c):
        print("  %-8s %8.1f M/s  frac %.3f  atomic %.3f  launch %.3f ms  U %.3f  traffic %s  early %.1f M/s  cpu %s  %s" % (n, v / 1e6, r["frac"], r["atomic_unit"]["frac"], r["avg_launch_ms"], r["updates_per_interaction"], r.get("traffic_over_algorithmic"), (e or {}).get("value", 0) / 1e6, (c or {}).get("value"), r["kernel"]))
    show("c2", d["value"], d["roofline"], d.get("early_epochs"), d.get("cpu_baseline"))
    for e in d.get("extra_configs", []):
        if "error" in e: print("  ", e)
        else: show(e["name"], e["value"], e["roofline"], e.get("early_epochs"), e.get("cpu_baseline"))
    q = d.get("quality") or {}
    print("  quality", q.get("precision_at_10"), q.get("precision_at_10_ref"), q.get("delta"))
    print("  cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("thread_scaling"), "fit", (d.get("end_to_end_fit") or {}).get("value"))
except Exception as e:
    print("  no result:", e)
PY
  tail -3 $OUT/bench_default.err | cut -c1-300
  ;;
r4e)
  # round 4: knob sweep of the row-stream kernel on the C5 shard (--scale 0.25; no code change): candidate batch,
  # wavefronts per workgroup, LDS budget / residency
  line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]
    print("  %-34s %9.2f M/s  frac %.3f  launch %.3f ms  S %.2f U %.3f  in_flight %s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["avg_launch_ms"], r["draws_per_interaction"], r["updates_per_interaction"], r.get("interactions_in_flight")))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S="--no-cpu-baseline --no-quality --no-fit"
  run() { tag=$1; shift; env $ENVV timeout 400 python bench.py $S "$@" > $OUT/$tag.json 2> $OUT/$tag.err; line $tag $OUT/$tag.json; }
  C5="--config c5shard --scale 0.25 --steps 2 --warmup 1 --epochs-per-step 1"
  ENVV= run c5_default $C5
  for fb in 3 5 7; do ENVV= run c5_fb$fb $C5 --first-batch $fb; done
  ENVV="LIGHTFM_AMD_FEAT_WAVES_PER_BLOCK=2" run c5_wpb2 $C5
  ENVV="LIGHTFM_AMD_FEAT_WAVES_PER_BLOCK=4" run c5_wpb4 $C5
  ENVV="LIGHTFM_AMD_FEAT_LDS_KB=22 LIGHTFM_AMD_FEAT_WAVES_PER_CU=7" run c5_lds22_w7 $C5
  ENVV="LIGHTFM_AMD_FEAT_LDS_KB=26 LIGHTFM_AMD_FEAT_WAVES_PER_CU=6" run c5_lds26_w6 $C5
  ENVV= run c5_default_2 $C5
  # k-OS: candidates' representations built in the same pass as the positives' (default) against the split gather
  ENVV="LIGHTFM_AMD_LIB=$R/lightfm_amd/_lib_split/liblfm_hip.so" run c5_split_gather $C5
  ENVV="LIGHTFM_AMD_LIB=$R/lightfm_amd/_lib_split/liblfm_hip.so" run c5_split_gather_2 $C5
  ENVV= run c5_default_3 $C5
  timeout -k 5 900 $PYT tests/test_hip_feat.py tests/test_hip_parity.py "tests/test_baseline_shapes.py::test_c3_shape_default_launch_plan_samples_exact" "tests/test_precision_parity.py::test_warp_identity_c2_regime_regularised" "tests/test_precision_parity.py::test_warp_kos_shared_tag_rows" tests/test_golden.py -m gpu -q -x -s > $OUT/tests.log 2>&1
  echo "tests: exit $?  $(grep -aE ' passed| failed' $OUT/tests.log | tail -1)"; summ $OUT/tests.log 12
  grep -a "per side, fixed" $OUT/tests.log | cut -c1-220
  ;;
r4f)
  # round 4: the fast (compile-time-offset, LDS-broadcast) reduce of the row-stream kernels against the previous build
  # (lightfm_amd/_lib_split = per-entry reduce), C5 shard --scale 0.25 and C3; exactness tests first
  timeout -k 5 900 $PYT tests/test_hip_feat.py tests/test_hip_parity.py tests/test_hip_round2.py tests/test_golden.py "tests/test_baseline_shapes.py::test_c3_shape_default_launch_plan_samples_exact" "tests/test_precision_parity.py::test_warp_kos_shared_tag_rows" "tests/test_precision_parity.py::test_bpr_tag_features_c3_regime" -m gpu -q -x > $OUT/tests.log 2>&1
  echo "tests: exit $?  $(grep -aE ' passed| failed' $OUT/tests.log | tail -1)"; summ $OUT/tests.log 12
  line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]
    print("  %-34s %9.2f M/s  frac %.3f  atomic %.3f  launch %.3f ms  S %.2f U %.3f  in_flight %s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["atomic_unit"]["frac"], r["avg_launch_ms"], r["draws_per_interaction"], r["updates_per_interaction"], r.get("interactions_in_flight")))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S="--no-cpu-baseline --no-quality --no-fit"
  run() { tag=$1; shift; env $ENVV timeout 400 python bench.py $S "$@" > $OUT/$tag.json 2> $OUT/$tag.err; line $tag $OUT/$tag.json; }
  PREV="LIGHTFM_AMD_LIB=$R/lightfm_amd/_lib_split/liblfm_hip.so"
  C5="--config c5shard --scale 0.25 --steps 2 --warmup 1 --epochs-per-step 1"
  C3="--config c3 --steps 3 --warmup 1 --epochs-per-step 2"
  for i in 1 2; do
    ENVV= run c5_fast_$i $C5
    ENVV=$PREV run c5_prev_$i $C5
  done
  ENVV= run c5_fast_fb3 $C5 --first-batch 3
  ENVV= run c5_fast_fb4 $C5 --first-batch 4
  ENVV="LIGHTFM_AMD_FEAT_WAVES_PER_CU=12 LIGHTFM_AMD_FEAT_LDS_KB=13" run c5_fast_w12 $C5
  for i in 1 2; do
    ENVV= run c3_fast_$i $C3
    ENVV=$PREV run c3_prev_$i $C3
  done
  ;;
r4g)
  # round 4: row-stream kernels after the fast reduce -- k-OS candidates reuse the positives' tile rows, 12 wavefronts per
  # CU for WARP / k-OS (default), residency sweep around it; exactness + precision gates first
  timeout -k 5 1200 $PYT tests/test_hip_feat.py tests/test_hip_parity.py tests/test_hip_round2.py tests/test_golden.py tests/test_lightfm_api.py "tests/test_baseline_shapes.py::test_c3_shape_default_launch_plan_samples_exact" "tests/test_precision_parity.py::test_warp_kos_shared_tag_rows" "tests/test_precision_parity.py::test_warp_shared_tag_rows" -m gpu -q -x -s > $OUT/tests.log 2>&1
  echo "tests: exit $?  $(grep -aE ' passed| failed' $OUT/tests.log | tail -1)"; summ $OUT/tests.log 12
  grep -a "per side, fixed" $OUT/tests.log | cut -c1-220
  line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]
    print("  %-34s %9.2f M/s  frac %.3f  atomic %.3f  launch %.3f ms  S %.2f U %.3f  in_flight %s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["atomic_unit"]["frac"], r["avg_launch_ms"], r["draws_per_interaction"], r["updates_per_interaction"], r.get("interactions_in_flight")))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S="--no-cpu-baseline --no-quality --no-fit"
  run() { tag=$1; shift; env $ENVV timeout 400 python bench.py $S "$@" > $OUT/$tag.json 2> $OUT/$tag.err; line $tag $OUT/$tag.json; }
  C5="--config c5shard --scale 0.25 --steps 2 --warmup 1 --epochs-per-step 1"
  ENVV= run c5_default_w12 $C5
  ENVV= run c5_default_w12_fb4 $C5 --first-batch 4
  ENVV="LIGHTFM_AMD_FEAT_WAVES_PER_CU=8 LIGHTFM_AMD_FEAT_LDS_KB=19" run c5_w8_lds19 $C5
  ENVV="LIGHTFM_AMD_FEAT_WAVES_PER_CU=10 LIGHTFM_AMD_FEAT_LDS_KB=15" run c5_w10_lds15 $C5
  ENVV="LIGHTFM_AMD_FEAT_WAVES_PER_CU=14 LIGHTFM_AMD_FEAT_LDS_KB=11" run c5_w14_lds11 $C5
  ENVV="LIGHTFM_AMD_FEAT_WAVES_PER_CU=16 LIGHTFM_AMD_FEAT_LDS_KB=10" run c5_w16_lds10 $C5
  ENVV="LIGHTFM_AMD_FEAT_WAVES_PER_CU=16 LIGHTFM_AMD_FEAT_LDS_KB=10" run c5_w16_lds10_fb4 $C5 --first-batch 4
  ENVV= run c5_default_w12_2 $C5
  ;;
r4h)
  # round 4: full GPU suite on the current tree, then row-stream kernels A/B against the previous commit's library
  # (lightfm_amd/_lib_prev): DMA address generation without ds_bpermute, four rows per issue step, empty-job clears only,
  # first batch of 4 candidates
  timeout -k 5 1500 $PYT tests -m gpu -x -q > $OUT/suite.log 2>&1
  echo "suite: exit $?  $(grep -aE ' passed| failed' $OUT/suite.log | tail -1)"; summ $OUT/suite.log 12
  line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]
    print("  %-34s %9.2f M/s  frac %.3f  atomic %.3f  launch %.3f ms  S %.2f U %.3f  in_flight %s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["atomic_unit"]["frac"], r["avg_launch_ms"], r["draws_per_interaction"], r["updates_per_interaction"], r.get("interactions_in_flight")))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S="--no-cpu-baseline --no-quality --no-fit"
  run() { tag=$1; shift; env $ENVV timeout 400 python bench.py $S "$@" > $OUT/$tag.json 2> $OUT/$tag.err; line $tag $OUT/$tag.json; }
  PREV="LIGHTFM_AMD_LIB=$R/lightfm_amd/_lib_prev/liblfm_hip.so"
  C5="--config c5shard --scale 0.25 --steps 2 --warmup 1 --epochs-per-step 1"
  C3="--config c3 --steps 3 --warmup 1 --epochs-per-step 2"
  for i in 1 2; do
    ENVV= run c5_new_$i $C5
    ENVV=$PREV run c5_prev_fb4_$i $C5 --first-batch 4
  done
  for i in 1 2; do
    ENVV= run c3_new_$i $C3
    ENVV=$PREV run c3_prev_$i $C3
  done
  ENVV= run c5_full_size --config c5shard --steps 2 --warmup 1 --epochs-per-step 1
  ;;
r4i)
  # round 4: instruction mix / issue utilisation of the steady-state tile kernel (C2) and the row-stream kernel (C3, C5): two PMC passes each
  cd /tmp && export TMPDIR=/tmp
  S="--no-cpu-baseline --no-quality --no-fit --steps 2 --warmup 1 --epochs-per-step 4"
  for cfg in "c2" "c3" "c5shard --scale 0.1"; do
    tag=$(echo $cfg | cut -d' ' -f1)
    EPS="--epochs-per-step 4"; [ "$tag" != "c2" ] && EPS="--epochs-per-step 1"
    S="--no-cpu-baseline --no-quality --no-fit --steps 2 --warmup 1 $EPS"
    timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/${tag}_a -o pmc -- python $R/bench.py $S --config $cfg > $OUT/${tag}_a.json 2> $OUT/${tag}_a.err
    timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU -d $OUT/${tag}_b -o pmc -- python $R/bench.py $S --config $cfg > $OUT/${tag}_b.json 2> $OUT/${tag}_b.err
  done
  cd $R && python - <<PY
import sqlite3, json, glob
for tag in ("c2", "c3", "c5shard"):
    for sub in ("a", "b"):
        try:
            db = glob.glob("$OUT/%s_%s/**/*results.db" % (tag, sub), recursive=True)[0]
            con = sqlite3.connect(db)
            rows = con.execute("select k.name, p.counter_name, sum(p.counter_value) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name").fetchall()
            b = json.load(open("$OUT/%s_%s.json" % (tag, sub)))
            tot = {}
            for name, c, v in rows:
                if "fit_" in name: tot[c] = tot.get(c, 0) + v
            eps = b["config"]["epochs_per_step"]; n_ep = 1 + 3 + max(0, 1 * eps - 4) + 2 * eps   # ramp + early + warm-up + timed
            n_int = float(b["config"]["workload"].split(" x ")[-1].split(" interactions")[0].replace(",", "")) * n_ep
            print("  %s_%s (%.1f M/s, %d epochs assumed):" % (tag, sub, b["value"] / 1e6, n_ep), {c: round(v / n_int, 1) for c, v in tot.items()})
        except Exception as e:
            print("  %s_%s: %r" % (tag, sub, e))
PY
  find $OUT -name "*.db" -size +5M -delete
  ;;
r4j)
  # round 4: the leaner steady-state tile kernel (lane-parallel bias cells, unconditional atomics, 32-bit gather offsets)
  # against the previous one (lightfm_amd/_lib_prev); exactness + gates first
  timeout -k 5 1200 $PYT tests/test_hip_warp_tile.py tests/test_hip_parity.py tests/test_baseline_shapes.py tests/test_sharded_items.py tests/test_hip_round2.py tests/test_lightfm_api.py tests/test_reference_suite.py "tests/test_precision_parity.py::test_warp_identity_c2_regime" -m gpu -q -x -s > $OUT/tests.log 2>&1
  echo "tests: exit $?  $(grep -aE ' passed| failed' $OUT/tests.log | tail -1)"; summ $OUT/tests.log 12
  grep -a "per side, fixed" $OUT/tests.log | cut -c1-220
  line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]
    print("  %-34s %9.2f M/s  frac %.3f  atomic %.3f  launch %.3f ms  S %.2f U %.3f  in_flight %s  early %.1f" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["atomic_unit"]["frac"], r["avg_launch_ms"], r["draws_per_interaction"], r["updates_per_interaction"], r.get("interactions_in_flight"), (d.get("early_epochs") or {}).get("value", 0) / 1e6))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S="--no-cpu-baseline --no-quality --no-fit"
  run() { tag=$1; shift; env $ENVV timeout 400 python bench.py $S "$@" > $OUT/$tag.json 2> $OUT/$tag.err; line $tag $OUT/$tag.json; }
  PREV="LIGHTFM_AMD_LIB=$R/lightfm_amd/_lib_prev/liblfm_hip.so"
  C2="--config c2 --steps 5 --warmup 2 --epochs-per-step 16"
  C4="--config c4shard --steps 3 --warmup 1 --epochs-per-step 8"
  for i in 1 2; do
    ENVV= run c2_new_$i $C2
    ENVV=$PREV run c2_prev_$i $C2
  done
  ENVV= run c4_new $C4
  ENVV=$PREV run c4_prev $C4
  ENVV= run c4_new_2 $C4
  ;;
r4k)
  # round 4: rocprofv3 kernel trace + FETCH / WRITE / TCC counter passes of the four configurations on the final kernels
  for cfg in c2 c3 c4shard; do bash tools/profile2.sh r04_$cfg --config $cfg; done
  bash tools/profile2.sh r04_c5shard --config c5shard --scale 0.1
  cp $R/profiles/r04_c*_kernel_stats.txt $R/profiles/r04_c*_pmc_summary.json $OUT/ 2>/dev/null
  ls -la $OUT | head -20
  ;;
r4m)
  # round 4: what predict_ranks is bound by -- kernel trace + SQ counters of tools/ranks_timing.py (all ML-20M users)
  cd /tmp && export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/tools/ranks_timing.py > $OUT/ranks_trace.txt 2> $OUT/ranks_trace.err
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc_a -o pmc -- python $R/tools/ranks_timing.py > $OUT/ranks_a.txt 2> $OUT/ranks_a.err
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU -d $OUT/pmc_b -o pmc -- python $R/tools/ranks_timing.py > $OUT/ranks_b.txt 2> $OUT/ranks_b.err
  cd $R && cat $OUT/ranks_trace.txt | tail -3 && python - <<PY
import sqlite3, glob
try:
    db = glob.glob("$OUT/trace/**/*results.db", recursive=True)[0]
    con = sqlite3.connect(db)
    for r in con.execute("select name, count(*), sum(duration)/1e6, avg(duration)/1e3 from kernels group by name order by sum(duration) desc limit 8"):
        print("  %-70s calls %4d total %9.2f ms avg %10.1f us" % (r[0][:70], r[1], r[2], r[3]))
except Exception as e:
    print("trace:", e)
for sub in ("pmc_a", "pmc_b"):
    try:
        db = glob.glob("$OUT/%s/**/*results.db" % sub, recursive=True)[0]
        con = sqlite3.connect(db)
        rows = con.execute("select k.name, p.counter_name, count(distinct p.dispatch_id), sum(p.counter_value) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id where k.name like '%ranks_mfma2%' group by k.name, p.counter_name").fetchall()
        print("  %s:" % sub, {c: round(v / n, 0) for _, c, n, v in rows}, "per launch")
    except Exception as e:
        print(sub, e)
PY
  find $OUT -name "*.db" -size +5M -delete
  ;;
r4n)
  # round 4: row-stream kernels with DMA row addresses through an LDS table (one ds_read_b64 per instruction, eight per
  # wait) against the previous commit's library; exactness + gates first
  timeout -k 5 1200 $PYT tests/test_hip_feat.py tests/test_hip_parity.py tests/test_hip_round2.py tests/test_golden.py "tests/test_baseline_shapes.py::test_c3_shape_default_launch_plan_samples_exact" "tests/test_precision_parity.py::test_warp_shared_tag_rows" "tests/test_precision_parity.py::test_bpr_tag_features_c3_regime" -m gpu -q -x -s > $OUT/tests.log 2>&1
  echo "tests: exit $?  $(grep -aE ' passed| failed' $OUT/tests.log | tail -1)"; summ $OUT/tests.log 12
  grep -a "per side, fixed" $OUT/tests.log | cut -c1-220
  line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]
    print("  %-34s %9.2f M/s  frac %.3f  atomic %.3f  launch %.3f ms  S %.2f U %.3f  in_flight %s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["atomic_unit"]["frac"], r["avg_launch_ms"], r["draws_per_interaction"], r["updates_per_interaction"], r.get("interactions_in_flight")))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S="--no-cpu-baseline --no-quality --no-fit"
  run() { tag=$1; shift; env $ENVV timeout 400 python bench.py $S "$@" > $OUT/$tag.json 2> $OUT/$tag.err; line $tag $OUT/$tag.json; }
  PREV="LIGHTFM_AMD_LIB=$R/lightfm_amd/_lib_prev/liblfm_hip.so"
  C5="--config c5shard --scale 0.25 --steps 2 --warmup 1 --epochs-per-step 1"
  C3="--config c3 --steps 3 --warmup 1 --epochs-per-step 2"
  for i in 1 2; do
    ENVV= run c5_new_$i $C5
    ENVV=$PREV run c5_prev_$i $C5
  done
  for i in 1 2; do
    ENVV= run c3_new_$i $C3
    ENVV=$PREV run c3_prev_$i $C3
  done
  ;;
r4o)
  # round 4: predict_ranks with the bucket-search sweep (ranks_mfma3_kernel) -- exactness first, then time against the
  # compare-chain sweep in one process, kernel trace + SQ counters
  timeout -k 5 900 $PYT tests/test_evaluation_gpu.py tests/test_golden.py tests/test_lightfm_api.py "tests/test_baseline_shapes.py::test_predict_ranks_vs_oracle_at_ml20m_items" tests/test_reference_suite.py -m gpu -q -x > $OUT/tests.log 2>&1
  echo "tests: exit $?  $(grep -aE ' passed| failed' $OUT/tests.log | tail -1)"; summ $OUT/tests.log 12
  cd /tmp && export TMPDIR=/tmp
  RANKS_TIMING_MODES=3,2 timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/tools/ranks_timing.py > $OUT/ranks_trace.txt 2> $OUT/ranks_trace.err
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc_a -o pmc -- python $R/tools/ranks_timing.py > $OUT/ranks_a.txt 2> $OUT/ranks_a.err
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU -d $OUT/pmc_b -o pmc -- python $R/tools/ranks_timing.py > $OUT/ranks_b.txt 2> $OUT/ranks_b.err
  cd $R && grep -a "mode\|identical" $OUT/ranks_trace.txt | cut -c1-330 && tail -3 $OUT/ranks_trace.err | cut -c1-300 && python - <<PY
import sqlite3, glob
try:
    db = glob.glob("$OUT/trace/**/*results.db", recursive=True)[0]
    con = sqlite3.connect(db)
    for r in con.execute("select name, count(*), sum(duration)/1e6, avg(duration)/1e3 from kernels group by name order by sum(duration) desc limit 8"):
        print("  %-70s calls %4d total %9.2f ms avg %10.1f us" % (r[0][:70], r[1], r[2], r[3]))
except Exception as e:
    print("trace:", e)
for sub in ("pmc_a", "pmc_b"):
    try:
        db = glob.glob("$OUT/%s/**/*results.db" % sub, recursive=True)[0]
        con = sqlite3.connect(db)
        rows = con.execute("select k.name, p.counter_name, count(distinct p.dispatch_id), sum(p.counter_value) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id where k.name like '%ranks_mfma3%' group by k.name, p.counter_name").fetchall()
        print("  %s:" % sub, {c: round(v / n, 0) for _, c, n, v in rows}, "per launch")
    except Exception as e:
        print(sub, e)
PY
  find $OUT -name "*.db" -size +5M -delete
  ;;
r4p)
  # round 4: predict_ranks stress -- the default kernel against the scalar kernel on random small dense problems
  timeout 600 python tools/ranks_stress.py ${1:-200} ${2:-0} > $OUT/stress.txt 2>&1
  tail -40 $OUT/stress.txt | cut -c1-400
  ;;
r4q)
  # round 4: where the bucket-search sweep's time goes -- exactness first, then timing-only variants of predict_kernels.hip
  # (tools/build_variants.sh: x1 no bucket atomics, x2 no LDS search steps, x3 no matrix products, x9 no exact re-checks,
  # lock = groups in step), then latency / wait counters of the shipped kernel
  timeout -k 5 600 $PYT tests/test_evaluation_gpu.py tests/test_golden.py tests/test_lightfm_api.py "tests/test_baseline_shapes.py::test_predict_ranks_vs_oracle_at_ml20m_items" -m gpu -q -x > $OUT/tests.log 2>&1
  echo "tests: exit $?  $(grep -aE ' passed| failed' $OUT/tests.log | tail -1)"; summ $OUT/tests.log 12
  timeout 300 python tools/ranks_stress.py 200 500 > $OUT/stress.txt 2>&1; grep -a "differ" $OUT/stress.txt | tail -3
  for v in _lib _lib_x9 _lib_lock _lib_x1 _lib_x2 _lib_x3 $*; do
    [ -f $R/lightfm_amd/$v/liblfm_hip.so ] || continue
    LIGHTFM_AMD_LIB=$R/lightfm_amd/$v/liblfm_hip.so RANKS_TIMING_MODES=4,3 timeout 200 python tools/ranks_timing.py > $OUT/t$v.txt 2>&1
    echo "$v: $(grep -a "mode [34]" $OUT/t$v.txt | sed -n "3p;6p" | sed 's/.*wall/wall/' | cut -c1-110)"
  done
  [ -n "$SKIP_PMC" ] && exit 0
  cd /tmp && export TMPDIR=/tmp
  i=0
  for set in "LdsLatency" "VmemLatency" "SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$i -o pmc -- python $R/tools/ranks_timing.py > $OUT/pmc_$i.txt 2> $OUT/pmc_$i.err
  done
  cd $R && python - <<PY
import sqlite3, glob
for i in (1, 2, 3, 4):
    try:
        db = glob.glob("$OUT/pmc_%d/**/*results.db" % i, recursive=True)[0]
        con = sqlite3.connect(db)
        rows = con.execute("select k.name, p.counter_name, count(distinct p.dispatch_id), sum(p.counter_value) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id where k.name like '%ranks_mfma3%' group by k.name, p.counter_name").fetchall()
        print("  pmc_%d:" % i, {c: round(v / n, 1) for _, c, n, v in rows}, "per launch")
        if i == 4:
            for r in con.execute("select name, count(*), avg(duration)/1e3 from kernels where name like '%ranks%' or name like '%test_scores%' or name like '%rep_rows%' or name like '%item_eps%' or name like '%rows_sorted%' group by name"):
                print("  %-60s calls %3d avg %9.1f us" % (r[0][:60], r[1], r[2]))
    except Exception as e:
        print("pmc_%d:" % i, e)
PY
  find $OUT -name "*.db" -size +5M -delete
  ;;
r4s)
  # round 4: predict_ranks at the other width classes (d = 128: two wavefronts per SIMD; d = 32), both sweeps
  for D in 128 32; do
    RANKS_TIMING_D=$D RANKS_TIMING_MODES=3,2 timeout 300 python tools/ranks_timing.py > $OUT/d$D.txt 2>&1
    grep -a "mode\|identical" $OUT/d$D.txt | sed 's/precision_at_k over.*wall/wall/' | cut -c1-200
  done
  ;;
r4t)
  # round 4: the rocprofv3 --kernel-trace --stats summary of a predict_ranks call (tools/ranks_timing.py, d = 64, the default kernel and
  # the compare-chain sweep in one process) as CSV, for profiles/
  cd /tmp && export TMPDIR=/tmp
  RANKS_TIMING_MODES=3,2 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o ranks -- python $R/tools/ranks_timing.py > $OUT/ranks_trace.txt 2> $OUT/ranks_trace.err
  cd $R; grep -a "mode\|identical" $OUT/ranks_trace.txt | sed 's/precision_at_k over.*wall/wall/' | cut -c1-200
  f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/ranks_kernel_stats.csv && head -12 $OUT/ranks_kernel_stats.csv | cut -c1-200
  find $OUT/trace -type f -size +2M -delete
  ;;
r4u)
  # round 4: predict_ranks -- exactness of the current build, then the segment count of the item table (no rebuild)
  timeout -k 5 600 $PYT tests/test_evaluation_gpu.py tests/test_golden.py "tests/test_baseline_shapes.py::test_predict_ranks_vs_oracle_at_ml20m_items" -m gpu -q -x > $OUT/tests.log 2>&1
  echo "tests: exit $?  $(grep -aE ' passed| failed' $OUT/tests.log | tail -1)"; summ $OUT/tests.log 12
  timeout 300 python tools/ranks_stress.py 150 300 > $OUT/stress.txt 2>&1; grep -a "differ" $OUT/stress.txt | tail -3
  for sg in ${*:-0}; do
    if [ "$sg" = 0 ]; then unset LIGHTFM_AMD_RANKS_SEGMENTS; else export LIGHTFM_AMD_RANKS_SEGMENTS=$sg; fi
    RANKS_TIMING_MODES=3 timeout 200 python tools/ranks_timing.py > $OUT/seg$sg.txt 2>&1
    echo "segments $sg: $(grep -a 'mode 3' $OUT/seg$sg.txt | tail -1 | sed 's/.*wall/wall/' | cut -c1-150)"
  done
  ;;
r4v)
  # round 4: the callers of predict_ranks outside tests/test_evaluation_gpu.py on the final tree
  timeout -k 5 600 $PYT tests/test_lightfm_api.py tests/test_reference_suite.py tests/test_abi.py -m gpu -q -x > $OUT/tests.log 2>&1
  echo "tests: exit $?  $(grep -aE ' passed| failed' $OUT/tests.log | tail -1)"; summ $OUT/tests.log 12
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  ;;
emu)
  # tools/visit.sh emu <shape> <epochs> <seeds> CONFIG...   (tools/multi_gpu_emulation.py on one GPU)
  SH=$1; EP=$2; SD=$3; shift 3
  EMU_SHAPE=$SH EMU_EPOCHS=$EP EMU_SEEDS=$SD timeout 1500 python tools/multi_gpu_emulation.py "$@" > $OUT/emu_${SH}_$$.txt 2>&1
  grep -a "K=" $OUT/emu_${SH}_$$.txt; tail -2 $OUT/emu_${SH}_$$.txt | grep -av "K=" | cut -c1-300
  ;;
ab)
  # A/B of two builds of the library on the same box: tools/visit.sh ab <dir under lightfm_amd/> [bench args]
  ALT=$1; shift
  S="--no-cpu-baseline --no-quality --no-fit --steps 10 --warmup 2 ${*:---config c2}"
  for i in 1 2; do for lib in _lib $ALT; do
    LIGHTFM_AMD_LIB=$R/lightfm_amd/$lib/liblfm_hip.so timeout 300 python bench.py $S > $OUT/bench_${lib}_$i.json 2> $OUT/bench_${lib}_$i.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_${lib}_$i.json")); r = d["roofline"]
    print("  %-14s run $i: %8.1f M/s  frac %.3f  launch %.3f ms  U %.3f" % ("$lib", d["value"] / 1e6, r["frac"], r["avg_launch_ms"], r["updates_per_interaction"]))
except Exception as e:
    print("  $lib run $i: no result:", e)
PY
  done; done
  ;;
r3d)
  # N-GPU semantics on one GPU: the sparse overlapped merge at the shipped intervals (C2, C3, scaled C4)
  export EMU_SEEDS=${EMU_SEEDS:-1,2,3}
  EMU_SHAPE=c2 timeout 900 python tools/multi_gpu_emulation.py 1:adagrad:4:16384:0 8:adagrad:4:16384:0:overlap 8:adagrad:4:16384:0:sparse > $OUT/emu_c2.txt 2>&1; grep -a "K=" $OUT/emu_c2.txt
  EMU_SHAPE=c3 EMU_EPOCHS=3 timeout 900 python tools/multi_gpu_emulation.py 1:adagrad:4:16384:0 8:adagrad:4:16384:0:overlap > $OUT/emu_c3.txt 2>&1; grep -a "K=" $OUT/emu_c3.txt
  EMU_SHAPE=c4s EMU_EPOCHS=3 EMU_SEEDS=1,2 timeout 1200 python tools/multi_gpu_emulation.py 1:adagrad:4:16384:0 8:adagrad:4:16384:0:overlap > $OUT/emu_c4s.txt 2>&1; grep -a "K=" $OUT/emu_c4s.txt
  tail -3 $OUT/emu_c4s.txt | cut -c1-300
  ;;
r5b)
  # round 5: multi-process owner-sharded item tables (HIP IPC, K processes on the one GPU), fused / all-rows / adadelta merges,
  # one-GPU merge overhead at the ML-20M shape
  timeout 900 $PYT tests/test_sharded_items_ipc.py tests/test_sharded_items.py tests/test_hip_round2.py tests/test_zz_rccl_comm.py -m gpu -q -x \
     -k "ipc or sharded or merge or communicator or rccl" > $OUT/tests.log 2>&1
  tail -15 $OUT/tests.log
  timeout 300 python3 tools/merge_overhead.py 12 > $OUT/merge_overhead.txt 2>&1; cat $OUT/merge_overhead.txt | tail -8
  ;;
r5e)
  # end-to-end fit stages after the native input scan; gather-ahead staleness A/B; overlapped-exchange study with
  # 10 seeds per arm; kernel traces + counters (incl. request sizes) of c2 and the C4 shard on the round-5 bench semantics
  LIGHTFM_AMD_TIMING=1 timeout 300 python3 - > $OUT/fit_timing.txt 2>&1 <<'PY'
import time, sys
sys.path.insert(0, ".")
from lightfm_amd import LightFM, synthetic
data = synthetic.named("ml-20m")
for rep in range(3):
    m = LightFM(no_components=64, loss="warp", random_state=3)
    t = time.perf_counter(); m.fit(data, epochs=10); dt = time.perf_counter() - t
    print("fit(10 epochs) %.1f ms = %.1f M interactions/s" % (1e3 * dt, data.nnz * 10 / dt / 1e6), flush=True)
m = LightFM(no_components=64, loss="warp", random_state=3)
t = time.perf_counter(); m.fit(data, epochs=20); dt = time.perf_counter() - t
print("fit(20 epochs) %.1f ms = %.1f M interactions/s" % (1e3 * dt, data.nnz * 20 / dt / 1e6), flush=True)
PY
  cat $OUT/fit_timing.txt | cut -c1-400
  timeout 600 python3 tools/ahead_staleness.py 12 10 > $OUT/ahead_staleness.txt 2>&1; grep -a "x" $OUT/ahead_staleness.txt | tail -12
  EMU_SHAPE=c2 EMU_SEEDS=1,2,3,4,5,6,7,8,9,10 timeout 900 python3 tools/multi_gpu_emulation.py 1:adagrad:4:16384:0 8:adagrad:4:16384:0:sparse \
     8:adagrad:4:16384:4194304:sparse 8:adagrad:4:16384:4194304:overlap 8:adagrad:4:16384:0:overlap > $OUT/emu_c2_10seeds.txt 2>&1
  grep -a "K=" $OUT/emu_c2_10seeds.txt
  bash tools/profile2.sh r05_c2 --config c2
  bash tools/profile2.sh r05_c4shard --config c4shard
  ;;
r5f)
  # kernel experiments, one box: user rows by plain stores (debug 2048) on c2 / c4shard, no bias snapshots on the C4 shard (debug 32),
  # the next epoch's shuffle written under the current epoch (on / off), the staleness A/B under the shipped ramp
  S="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 2 --steady-seconds 3 --steps 20 --warmup 5"
  line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]; ss = d["config"].get("steady_state", {})
    print("  %-34s %8.1f M/s (fits %s)  frac %.3f  launch %.3f ms  U %.3f | steady %8.1f M/s frac %.3f" % (
        sys.argv[1], d["value"] / 1e6, [round(x / 1e6) for x in d["config"].get("fresh_fits", [])], r["frac"], r["avg_launch_ms"],
        r["updates_per_interaction"], ss.get("value", 0) / 1e6, ss.get("kernel_frac", 0)))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  for rep in 1 2; do
    for arm in "c2:default:" "c2:ustore:--debug 2048" "c2:noahead:" ; do
      IFS=: read cfg name extra <<< "$arm"
      if [ "$name" = noahead ]; then export LIGHTFM_AMD_SHUFFLE_AHEAD=0; else export LIGHTFM_AMD_SHUFFLE_AHEAD=1; fi
      timeout 300 python3 bench.py $S --config $cfg $extra > $OUT/${cfg}_${name}_$rep.json 2> $OUT/${cfg}_${name}_$rep.err
      line "$cfg $name run $rep" $OUT/${cfg}_${name}_$rep.json
    done
  done
  export LIGHTFM_AMD_SHUFFLE_AHEAD=1
  S4="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 2 --steps 8 --warmup 3"
  for rep in 1 2; do
    for arm in "c4shard:default:" "c4shard:nosnap:--debug 32" "c4shard:ustore:--debug 2048" "c4shard:both:--debug 2080"; do
      IFS=: read cfg name extra <<< "$arm"
      timeout 300 python3 bench.py $S4 --config $cfg $extra > $OUT/${cfg}_${name}_$rep.json 2> $OUT/${cfg}_${name}_$rep.err
      line "$cfg $name run $rep" $OUT/${cfg}_${name}_$rep.json
    done
  done
  QUALITY_DEBUG=0 timeout 600 python3 tools/quality20m.py 3 20000 1,2,3,4,5,6,7,8 hip0 > $OUT/quality_default.txt 2>&1; tail -1 $OUT/quality_default.txt
  QUALITY_DEBUG=2048 timeout 600 python3 tools/quality20m.py 3 20000 1,2,3,4,5,6,7,8 hip0 > $OUT/quality_ustore.txt 2>&1; tail -1 $OUT/quality_ustore.txt
  AHEAD_RAMP=1 timeout 600 python3 tools/ahead_staleness.py 12 10 > $OUT/ahead_staleness_ramp.txt 2>&1; grep -a " x " $OUT/ahead_staleness_ramp.txt | tail -12
  ;;
r5i)
  # k-OS row-stream kernel held to 128 VGPRs (build -DLFM_KOS_WAVES4 -> lightfm_amd/_lib_kos4): 16 wavefronts per CU with a
  # 10 KB LDS budget per wavefront (6 staged rows) against the shipped 12; C5 shard at --scale 0.25
  S="--config c5shard --scale 0.25 --no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 0 --steps 3 --warmup 1"
  run() { name=$1; shift; env "$@" timeout 300 python3 bench.py $S > $OUT/$name.json 2> $OUT/$name.err
    python3 - "$name" $OUT/$name.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]
    print("  %-28s %7.2f M/s  frac %.3f  launch %.2f ms  in flight %d  kernel %s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["avg_launch_ms"], r["interactions_in_flight"], r["kernel"]))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  for rep in 1 2; do
    run default12_$rep LIGHTFM_AMD_LIB=$R/lightfm_amd/_lib/liblfm_hip.so
    run kos4_12_$rep LIGHTFM_AMD_LIB=$R/lightfm_amd/_lib_kos4/liblfm_hip.so
    run kos4_16_lds10_$rep LIGHTFM_AMD_LIB=$R/lightfm_amd/_lib_kos4/liblfm_hip.so LIGHTFM_AMD_FEAT_WAVES_PER_CU=16 LIGHTFM_AMD_FEAT_LDS_KB=10
    run kos4_14_lds11_$rep LIGHTFM_AMD_LIB=$R/lightfm_amd/_lib_kos4/liblfm_hip.so LIGHTFM_AMD_FEAT_WAVES_PER_CU=14 LIGHTFM_AMD_FEAT_LDS_KB=11
  done
  ;;
r5j)
  # end-to-end fit with the native initialisation; kernel traces + counter passes of c2 (plain-store user rows), c3 and the C5 shard
  LIGHTFM_AMD_TIMING=1 timeout 300 python3 - > $OUT/fit_timing.txt 2>&1 <<'PY'
import time, sys
sys.path.insert(0, ".")
from lightfm_amd import LightFM, synthetic
data = synthetic.named("ml-20m")
for rep in range(4):
    m = LightFM(no_components=64, loss="warp", random_state=3)
    t = time.perf_counter(); m.fit(data, epochs=10); dt = time.perf_counter() - t
    print("fit(10 epochs) %.1f ms = %.1f M interactions/s" % (1e3 * dt, data.nnz * 10 / dt / 1e6), flush=True)
PY
  cat $OUT/fit_timing.txt | cut -c1-330
  bash tools/profile2.sh r05_c2 --config c2
  PROF_STEPS=5 PROF_WARMUP=2 bash tools/profile2.sh r05_c3 --config c3
  PROF_STEPS=3 PROF_WARMUP=1 bash tools/profile2.sh r05_c5shard --config c5shard --scale 0.25
  ;;
r5z)
  # the driver's sequence on the final tree: GPU suite, smoke, default bench; then the C4 shard's trace + counters on the shipped defaults
  timeout 1200 $PYT tests -m gpu -x -q > $OUT/suite.txt 2>&1; tail -3 $OUT/suite.txt
  timeout 300 python3 __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $? bytes $(wc -c < $OUT/bench.json)"
  grep -a "failed" $OUT/bench.err
  bash tools/profile2.sh r05_c4shard --config c4shard
  ;;
r5l)
  # launch length of the tile kernel (2 Mi shipped; 4 Mi / 1 Mi) on c2, residency of the BPR row-stream kernel on c3 (8 shipped; 6 / 10 / 12)
  line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]; ss = d["config"].get("steady_state", {})
    print("  %-26s %8.1f M/s  frac %.3f  launch %.3f ms  in flight %d | steady %8.1f M/s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["avg_launch_ms"], r["interactions_in_flight"], ss.get("value", 0) / 1e6))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 2 --steady-seconds 2 --steps 20 --warmup 5 --config c2"
  for rep in 1 2; do for lg in 21 22 20; do
    LIGHTFM_AMD_LAUNCH_LOG2=$lg timeout 200 python3 bench.py $S > $OUT/c2_log2_${lg}_$rep.json 2> $OUT/c2_log2_${lg}_$rep.err; line "c2 launch 2^$lg run $rep" $OUT/c2_log2_${lg}_$rep.json
  done; done
  S3="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5 --steps 4 --warmup 2 --config c3"
  for rep in 1 2; do for wv in 8 6 10 12; do
    LIGHTFM_AMD_FEAT_WAVES_PER_CU=$wv timeout 200 python3 bench.py $S3 > $OUT/c3_waves_${wv}_$rep.json 2> $OUT/c3_waves_${wv}_$rep.err; line "c3 waves/CU $wv run $rep" $OUT/c3_waves_${wv}_$rep.json
  done; done
  ;;
r6a)
  # round 6, first visit: the new parity tests at the HBM-bound shapes, the small-shape gates, the LDS read-modify-write rate,
  # and C2 / C3 short bench runs (user rows by plain stores must still be chosen on C2 under the collision-rate rule)
  free -g | head -2; nproc
  timeout 200 tools/_bin/membench lds > $OUT/membench_lds.txt 2>&1; echo "membench rc $?"; cat $OUT/membench_lds.txt
  ( time timeout 1500 $PYT tests/test_hbm_shapes.py -m gpu -x -q -s ) > $OUT/hbm_shapes.txt 2>&1; tail -5 $OUT/hbm_shapes.txt; grep real $OUT/hbm_shapes.txt
  ( time timeout 1500 $PYT tests/test_precision_parity.py -m gpu -q -s -k "ml100k or tiny" ) > $OUT/small_gates.txt 2>&1; grep -aE "delta|passed|failed|real" $OUT/small_gates.txt
  S="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 2 --steps 20 --warmup 5"
  timeout 300 python3 bench.py $S --config c2 > $OUT/c2.json 2> $OUT/c2.err; echo "c2 rc $?"; python3 -c "
import json; d=json.load(open('$OUT/c2.json')); print(d['value']/1e6, d['roofline']['frac'], d['roofline']['kernel'])"
  ;;
r6b)
  # which cells change in a frozen-weight epoch over an item table beyond 4 GB (tests/test_hbm_shapes.py failure of r6a)
  for args in "20000000 0" "20000000 1024" "20000000 0 noepoch" "16000000 0"; do
    timeout 300 python3 tools/debug_big_items.py $args 2>&1 | tail -8
  done
  ;;
r6c)
  # the N > 1 merge path over the stand-in collective (tests/fake_rccl.hip): K = 2, 4 rank processes on the one GPU + bench.py --gpus 2;
  # the HBM-shape parity tests again
  ( time timeout 1700 $PYT tests/test_fake_rccl_multirank.py -m gpu -x -q -s ) > $OUT/fake_rccl.txt 2>&1; grep -aE "FAKE_RCCL case|passed|failed|real|Error|error|assert" $OUT/fake_rccl.txt | cut -c1-300 | head -60
  ( time timeout 1500 $PYT tests/test_hbm_shapes.py -m gpu -x -q -s ) > $OUT/hbm_shapes.txt 2>&1; tail -5 $OUT/hbm_shapes.txt | cut -c1-300; grep real $OUT/hbm_shapes.txt
  ;;
r6d)
  # hot slices (csrc/hot_slices.hip): parity tests, the feature-kernel suites, then C3 A/B: hot set on / off, chunk and replica sweeps
  ( time timeout 1200 $PYT tests/test_hot_slices.py tests/test_hip_feat.py -m gpu -x -q ) > $OUT/tests.txt 2>&1; tail -15 $OUT/tests.txt | cut -c1-300
  line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]; ss = d["config"].get("steady_state", {})
    print("  %-34s %8.2f M/s  frac %.3f  launch %.3f ms  in flight %d | steady %8.2f M/s  %s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["avg_launch_ms"], r["interactions_in_flight"], ss.get("value", 0) / 1e6, r["kernel"][:60]))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S3="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5 --steps 4 --warmup 2 --config c3"
  for arm in "off:LIGHTFM_AMD_HOT_SLICES=0" "on:X=1" "chunk16k:LIGHTFM_AMD_HOT_CHUNK=16384" "chunk64k:LIGHTFM_AMD_HOT_CHUNK=65536" "chunk128k:LIGHTFM_AMD_HOT_CHUNK=131072" \
             "rep15:LIGHTFM_AMD_HOT_REPLICAS=15" "rep60:LIGHTFM_AMD_HOT_REPLICAS=60" "waves12:LIGHTFM_AMD_FEAT_WAVES_PER_CU=12" "waves16:LIGHTFM_AMD_FEAT_WAVES_PER_CU=16"; do
    IFS=: read name envs <<< "$arm"
    env $envs timeout 300 python3 bench.py $S3 > $OUT/c3_$name.json 2> $OUT/c3_$name.err; line "c3 $name" $OUT/c3_$name.json
  done
  ;;
r6e)
  # hot slices: quality against the record length -- full-size C3 precision@10 (this backend only; the reference: 0.06049, profiles/r05_quality_c3_full.txt)
  # and the hybrid precision gates at the default and at a 128 Ki record length
  for envs in "LIGHTFM_AMD_HOT_SLICES=0" "LIGHTFM_AMD_HOT_CHUNK=32768" "LIGHTFM_AMD_HOT_CHUNK=65536" "LIGHTFM_AMD_HOT_CHUNK=131072" "LIGHTFM_AMD_HOT_CHUNK=262144" "LIGHTFM_AMD_HOT_CHUNK=131072 LIGHTFM_AMD_HOT_K=2"; do
    env $envs timeout 400 python3 tools/quality_c3_full.py 0 2>&1 | tail -1
  done
  for envs in "LIGHTFM_AMD_HOT_CHUNK=32768" "LIGHTFM_AMD_HOT_CHUNK=131072"; do
    ( time env $envs timeout 1500 $PYT tests/test_precision_parity.py -m gpu -q -s -k "bpr_tag or shared_tag" ) > $OUT/gates_$(echo $envs | tr -c 'A-Za-z0-9' '_').txt 2>&1
    echo "== $envs"; grep -aE "delta|passed|failed|real" $OUT/gates_$(echo $envs | tr -c 'A-Za-z0-9' '_').txt
  done
  ;;
r6f)
  # hot slices: the record-length ramp (hot_k) and cap against the hybrid gate problems; then the feature suites with the hot set and the
  # adadelta instantiations
  timeout 2400 python3 tools/hot_gate_sweep.py 8 "off:HOT_SLICES=0" "k8-32k:HOT_K=8,HOT_CHUNK=32768" "k32-128k:HOT_K=32,HOT_CHUNK=131072" "k64-128k:HOT_K=64,HOT_CHUNK=131072" \
      "k128-128k:HOT_K=128,HOT_CHUNK=131072" "k256-128k:HOT_K=256,HOT_CHUNK=131072" "k64-32k:HOT_K=64,HOT_CHUNK=32768" 2>&1 | tail -30
  ( time timeout 1500 $PYT tests/test_hot_slices.py tests/test_hip_feat.py -m gpu -x -q ) > $OUT/tests.txt 2>&1; tail -6 $OUT/tests.txt | cut -c1-300
  ;;
r6g)
  # hot slices after the all-or-none rule and with the in-flight bound restored: suites, the gate sweep, C3 at the new defaults + profile
  ( time timeout 1500 $PYT tests/test_hot_slices.py tests/test_hip_feat.py -m gpu -x -q ) > $OUT/tests.txt 2>&1; tail -4 $OUT/tests.txt | cut -c1-300
  timeout 2400 python3 tools/hot_gate_sweep.py 8 "off:HOT_SLICES=0" "default-k128-128k:X=1" "k128-32k:HOT_K=128,HOT_CHUNK=32768" "k64-128k:HOT_K=64,HOT_CHUNK=131072" "k256-128k:HOT_K=256,HOT_CHUNK=131072" 2>&1 | tail -20
  S3="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5 --steps 5 --warmup 2 --config c3"
  timeout 300 python3 bench.py $S3 > $OUT/c3.json 2> $OUT/c3.err; python3 -c "
import json; d=json.load(open('$OUT/c3.json')); r=d['roofline']; print('c3 %.2f M/s frac %.3f steady %.2f  %s' % (d['value']/1e6, r['frac'], d['config'].get('steady_state',{}).get('value',0)/1e6, r['kernel']))"
  TRACE_ONLY=1 PROF_STEPS=5 PROF_WARMUP=2 bash tools/profile2.sh r06_c3 --config c3
  head -12 $R/profiles/r06_c3_kernel_stats.txt | cut -c1-200
  ;;
r6h)
  # hot slices: the floor of the record-length ramp (8 = the in-flight ramp's own start) on the WARP / k-OS hybrid gates
  SWEEP_PROBLEMS=warp-200tags-d64,kos-100tags-d64 timeout 2400 python3 tools/hot_gate_sweep.py 8 "off:HOT_SLICES=0" "floor8-k128:X=1" "floor8-k32:HOT_K=32" "floor64-k128:HOT_FLOOR=64" "floor256-k128:HOT_FLOOR=256" "floor8-k128-rep8:HOT_REPLICAS=8" 2>&1 | tail -20
  S3="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5 --steps 5 --warmup 2 --config c3"
  timeout 300 python3 bench.py $S3 > $OUT/c3.json 2> $OUT/c3.err; python3 -c "
import json; d=json.load(open('$OUT/c3.json')); r=d['roofline']; print('c3 %.2f M/s frac %.3f steady %.2f  %s' % (d['value']/1e6, r['frac'], d['config'].get('steady_state',{}).get('value',0)/1e6, r['kernel']))"
  ;;
r6i)
  # the adagrad cell without float64 root / quotient: self-test + hot suites, C3 with it (hot_slice_kernel), then the whole GPU suite
  ( time timeout 1500 $PYT tests/test_hot_slices.py -m gpu -x -q -s ) > $OUT/hot.txt 2>&1; grep -aE "mismatches|passed|failed|real" $OUT/hot.txt
  S3="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5 --steps 5 --warmup 2 --config c3"
  timeout 300 python3 bench.py $S3 > $OUT/c3.json 2> $OUT/c3.err; python3 -c "
import json; d=json.load(open('$OUT/c3.json')); r=d['roofline']; print('c3 %.2f M/s frac %.3f steady %.2f  %s' % (d['value']/1e6, r['frac'], d['config'].get('steady_state',{}).get('value',0)/1e6, r['kernel']))"
  TRACE_ONLY=1 PROF_STEPS=5 PROF_WARMUP=2 bash tools/profile2.sh r06_c3 --config c3
  head -6 $R/profiles/r06_c3_kernel_stats.txt | cut -c1-200
  ( time timeout 2400 $PYT tests -m gpu -x -q ) > $OUT/suite.txt 2>&1; tail -5 $OUT/suite.txt | cut -c1-300
  ;;
r6j)
  # the adagrad cell without float64 root / quotient in the tile and row-stream kernels: exactness suites, then A/B against the build
  # before it (lightfm_amd/_lib_before: the hot-slice kernel has it in both) on c2 / c4shard / c3 / c5shard
  ( time timeout 1800 $PYT tests/test_hip_warp_tile.py tests/test_hip_feat.py tests/test_hot_slices.py tests/test_baseline_shapes.py tests/test_evaluation_gpu.py -m gpu -x -q ) > $OUT/tests.txt 2>&1; tail -4 $OUT/tests.txt | cut -c1-300
  S="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5"
  for i in 1 2; do for cfg in "c2 --steps 20 --warmup 5" "c4shard --steps 6 --warmup 2" "c3 --steps 4 --warmup 2" "c5shard --steps 2 --warmup 1 --scale 0.25"; do for lib in _lib _lib_before; do
    name=$(echo $cfg | cut -d' ' -f1)
    LIGHTFM_AMD_LIB=$R/lightfm_amd/$lib/liblfm_hip.so timeout 400 python3 bench.py $S --config $cfg > $OUT/${name}_${lib}_$i.json 2> $OUT/${name}_${lib}_$i.err
    python3 - <<PY
import json
try:
    d = json.load(open("$OUT/${name}_${lib}_$i.json")); r = d["roofline"]; ss = d["config"].get("steady_state", {})
    print("  %-8s %-12s run $i: %8.2f M/s  frac %.3f  launch %.3f ms | steady %8.2f M/s" % ("$name", "$lib", d["value"] / 1e6, r["frac"], r["avg_launch_ms"], ss.get("value", 0) / 1e6))
except Exception as e:
    print("  $name $lib run $i: no result:", e)
PY
  done; done; done
  ;;
r6k)
  # the slice kernel of launch k under launch k + 1 (second stream): suites, C3 with / without (same box), the hybrid gates with it
  ( time timeout 1500 $PYT tests/test_hot_slices.py tests/test_hip_feat.py -m gpu -x -q ) > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt | cut -c1-300
  line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]; ss = d["config"].get("steady_state", {})
    print("  %-30s %8.2f M/s  frac %.3f  launch %.3f ms  in flight %d | steady %8.2f M/s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["avg_launch_ms"], r["interactions_in_flight"], ss.get("value", 0) / 1e6))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S3="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5 --steps 5 --warmup 2 --config c3"
  for i in 1 2; do for arm in "overlap:X=1" "inline:LIGHTFM_AMD_HOT_OVERLAP=0" "overlap-k256:LIGHTFM_AMD_HOT_K=256" "overlap-64k:LIGHTFM_AMD_HOT_CHUNK=65536" "overlap-waves12:LIGHTFM_AMD_FEAT_WAVES_PER_CU=12"; do
    IFS=: read name envs <<< "$arm"
    env $envs timeout 300 python3 bench.py $S3 > $OUT/c3_${name}_$i.json 2> $OUT/c3_${name}_$i.err; line "c3 $name run $i" $OUT/c3_${name}_$i.json
  done; done
  timeout 2400 python3 tools/hot_gate_sweep.py 8 "off:HOT_SLICES=0" "inline:HOT_OVERLAP=0" "overlap:X=1" "overlap-k256:HOT_K=256" 2>&1 | tail -16
  ;;
r6l)
  # C3: 12 row-stream wavefronts per CU with the hot set, BPR candidate / positives-range prefetch; slice-kernel workgroup size; then C5 and the suites
  ( time timeout 1500 $PYT tests/test_hot_slices.py tests/test_hip_feat.py tests/test_baseline_shapes.py -m gpu -x -q ) > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt | cut -c1-300
  line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]; ss = d["config"].get("steady_state", {})
    print("  %-30s %8.2f M/s  frac %.3f  launch %.3f ms  in flight %d | steady %8.2f M/s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["avg_launch_ms"], r["interactions_in_flight"], ss.get("value", 0) / 1e6))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S3="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5 --steps 5 --warmup 2 --config c3"
  for i in 1 2; do for arm in "default:X=1" "threads1024:LIGHTFM_AMD_HOT_THREADS=1024" "threads256:LIGHTFM_AMD_HOT_THREADS=256" "rep45:LIGHTFM_AMD_HOT_REPLICAS=45" "chunk256k:LIGHTFM_AMD_HOT_CHUNK=262144" "off:LIGHTFM_AMD_HOT_SLICES=0"; do
    IFS=: read name envs <<< "$arm"
    env $envs timeout 300 python3 bench.py $S3 > $OUT/c3_${name}_$i.json 2> $OUT/c3_${name}_$i.err; line "c3 $name run $i" $OUT/c3_${name}_$i.json
  done; done
  TRACE_ONLY=1 PROF_STEPS=5 PROF_WARMUP=2 bash tools/profile2.sh r06_c3 --config c3
  head -6 $R/profiles/r06_c3_kernel_stats.txt | cut -c1-200
  ;;
r6m)
  # slice kernel: records prefetched two deep (_lib) against one deep (_lib_d1), both with 1 024-thread workgroups; record length 128 / 256 Ki;
  # then the hot suites and the hybrid gates on the shipped defaults, full-size C3 quality at both record lengths
  line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]; ss = d["config"].get("steady_state", {})
    print("  %-30s %8.2f M/s  frac %.3f  launch %.3f ms  in flight %d | steady %8.2f M/s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["avg_launch_ms"], r["interactions_in_flight"], ss.get("value", 0) / 1e6))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S3="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5 --steps 5 --warmup 2 --config c3"
  for i in 1 2; do for arm in "depth2:_lib:X=1" "depth1:_lib_d1:X=1" "depth2-256k:_lib:LIGHTFM_AMD_HOT_CHUNK=262144" "depth2-rep20:_lib:LIGHTFM_AMD_HOT_REPLICAS=20"; do
    IFS=: read name lib envs <<< "$arm"
    env $envs LIGHTFM_AMD_LIB=$R/lightfm_amd/$lib/liblfm_hip.so timeout 300 python3 bench.py $S3 > $OUT/c3_${name}_$i.json 2> $OUT/c3_${name}_$i.err; line "c3 $name run $i" $OUT/c3_${name}_$i.json
  done; done
  ( time timeout 1500 $PYT tests/test_hot_slices.py tests/test_hip_feat.py -m gpu -x -q ) > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt | cut -c1-300
  timeout 2400 python3 tools/hot_gate_sweep.py 8 "off:HOT_SLICES=0" "shipped:X=1" 2>&1 | tail -10
  for envs in "LIGHTFM_AMD_HOT_CHUNK=131072" "LIGHTFM_AMD_HOT_CHUNK=262144"; do
    env $envs timeout 400 python3 tools/quality_c3_full.py 0 2>&1 | tail -1
  done
  ;;
r6n)
  # the steady-state tile kernel with rows of <= 16 floats (VEC = 1: the reference's default width): parity suites, then c2 at d = 10 / 16 --
  # 64-float layout (LIGHTFM_AMD_TILE_NARROW=0), VEC = 1 compiled for 4 (_lib) and 5 (_lib_nb5) workgroups per CU, with / without the plain-store
  # user rows; c5shard with / without the next-position bounds prefetch (_lib_nopre)
  ( time timeout 1800 $PYT tests/test_hip_warp_tile.py tests/test_hip_parity.py tests/test_baseline_shapes.py tests/test_sharded_items.py tests/test_hip_feat.py tests/test_hot_slices.py -m gpu -x -q ) > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt | cut -c1-300
  line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]; ss = d["config"].get("steady_state", {})
    print("  %-34s %8.2f M/s  frac %.3f  launch %.3f ms  in flight %d  ustore %s | steady %8.2f M/s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["avg_launch_ms"], r["interactions_in_flight"], r.get("user_rows_by_plain_stores"), ss.get("value", 0) / 1e6))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5 --steps 10 --warmup 3 --config c2"
  for i in 1 2; do for arm in "d10-wide:_lib:LIGHTFM_AMD_TILE_NARROW=0:10" "d10-narrow4:_lib:X=1:10" "d10-narrow5:_lib_nb5:X=1:10" "d10-narrow4-noustore:_lib:X=1:10:--debug 4096" "d10-narrow5-noustore:_lib_nb5:X=1:10:--debug 4096" "d16-wide:_lib:LIGHTFM_AMD_TILE_NARROW=0:16" "d16-narrow4:_lib:X=1:16"; do
    IFS=: read name lib envs dd extra <<< "$arm"
    env $envs LIGHTFM_AMD_LIB=$R/lightfm_amd/$lib/liblfm_hip.so timeout 300 python3 bench.py $S --no-components $dd $extra > $OUT/c2_${name}_$i.json 2> $OUT/c2_${name}_$i.err; line "c2 $name run $i" $OUT/c2_${name}_$i.json
  done; done
  S5="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5 --steps 2 --warmup 1 --config c5shard --scale 0.25"
  for i in 1 2; do for lib in _lib _lib_nopre; do
    LIGHTFM_AMD_LIB=$R/lightfm_amd/$lib/liblfm_hip.so timeout 400 python3 bench.py $S5 > $OUT/c5_${lib}_$i.json 2> $OUT/c5_${lib}_$i.err; line "c5shard $lib run $i" $OUT/c5_${lib}_$i.json
  done; done
  ;;
r6o)
  # user bias cells by plain stores with the user rows (_lib) against by atomics (_lib_prev): tile suites, c2 at d = 64 and d = 10
  ( time timeout 1800 $PYT tests/test_hip_warp_tile.py tests/test_baseline_shapes.py -m gpu -x -q ) > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt | cut -c1-300
  line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]; ss = d["config"].get("steady_state", {})
    print("  %-34s %8.2f M/s  frac %.3f  launch %.3f ms  in flight %d  ustore %s | steady %8.2f M/s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["avg_launch_ms"], r["interactions_in_flight"], r.get("user_rows_by_plain_stores"), ss.get("value", 0) / 1e6))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5 --steps 15 --warmup 5 --config c2"
  for i in 1 2 3; do for arm in "d64:_lib:64" "d64-prev:_lib_prev:64" "d10:_lib:10" "d10-prev:_lib_prev:10"; do
    IFS=: read name lib dd <<< "$arm"
    LIGHTFM_AMD_LIB=$R/lightfm_amd/$lib/liblfm_hip.so timeout 300 python3 bench.py $S --no-components $dd > $OUT/c2_${name}_$i.json 2> $OUT/c2_${name}_$i.err; line "c2 $name run $i" $OUT/c2_${name}_$i.json
  done; done
  ;;
r6p)
  # kernel traces + counter passes of the four configurations on the round's final kernels
  bash tools/profile2.sh r06_c2 --config c2
  PROF_STEPS=5 PROF_WARMUP=2 bash tools/profile2.sh r06_c3 --config c3
  PROF_STEPS=8 PROF_WARMUP=3 bash tools/profile2.sh r06_c4shard --config c4shard
  PROF_STEPS=3 PROF_WARMUP=1 bash tools/profile2.sh r06_c5shard --config c5shard --scale 0.25
  ls -la $R/gpurun_out/prof_r06_*/bench_fetch.json $R/gpurun_out/prof_r06_*/bench_trace.json 2>/dev/null | head
  ;;
r6p2)
  # the final tree's traces: C2 and the C4 shard with all counter passes (both run the steady-state kernel with bias pairs now), kernel traces
  # of the narrow and identity-model kernels (c2 at d = 10, LightFM()'s default model, BPR at d = 64)
  bash tools/profile2.sh r06_c2 --config c2
  PROF_STEPS=8 PROF_WARMUP=3 bash tools/profile2.sh r06_c4shard --config c4shard
  TRACE_ONLY=1 PROF_STEPS=8 PROF_WARMUP=3 bash tools/profile2.sh r06_c2_d10 --config c2 --no-components 10
  TRACE_ONLY=1 PROF_STEPS=8 PROF_WARMUP=3 bash tools/profile2.sh r06_default_model --config default_model
  TRACE_ONLY=1 PROF_STEPS=8 PROF_WARMUP=3 bash tools/profile2.sh r06_c2_bpr --config c2_bpr
  ;;
r6q)
  # multi-GPU semantics with the hot set in every emulated rank: C3, K = 1 against K = 8 (sparse merges + hot rows at their cadence), 3 seeds;
  # C2's per-rank kernel time at N = 8 (rank 0's shard of the strong-scaling split on this one GPU)
  EMU_SHAPE=c3 EMU_EPOCHS=3 timeout 1500 python3 tools/multi_gpu_emulation.py 1:adagrad:4:16384:0 8:adagrad:4:16384:0:sparse+hot > $OUT/emu_c3.txt 2>&1; grep -a "K=" $OUT/emu_c3.txt
  timeout 400 python3 bench.py --config c2 --emulate-shard 8 --no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1 --steps 20 --warmup 5 > $OUT/c2_shard8.json 2> $OUT/c2_shard8.err
  python3 -c "
import json; d=json.load(open('$OUT/c2_shard8.json')); r=d['roofline']; print('c2 rank-0 shard of 8: %.1f M/s, %.3f ms per epoch, ustore %s, in flight %d' % (d['value']/1e6, d['ms_per_step'], r.get('user_rows_by_plain_stores'), r['interactions_in_flight']))"
  ;;
r6r)
  # the hot set taken only once a launch reaches 2 048 positions: suites, gates, C3's first epochs (LightFM.fit of 3 epochs, wall time per epoch)
  ( time timeout 1500 $PYT tests/test_hot_slices.py tests/test_hip_feat.py -m gpu -x -q ) > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt | cut -c1-300
  for envs in "LIGHTFM_AMD_HOT_MIN=2048" "LIGHTFM_AMD_HOT_MIN=1" "LIGHTFM_AMD_HOT_SLICES=0"; do
    env $envs LIGHTFM_AMD_TIMING=1 timeout 400 python3 - 2>&1 <<'PY' | grep -aE "timing|fit\(" | cut -c1-400
import os, sys, time
sys.path.insert(0, ".")
from lightfm_amd import LightFM, synthetic
data = synthetic.named("ml-20m")
feats = synthetic.tag_item_features(data.shape[1])
for rep in range(2):
    m = LightFM(no_components=128, loss="bpr", random_state=3)
    t = time.perf_counter(); m.fit(data, item_features=feats, epochs=3); dt = time.perf_counter() - t
    print("fit(3 epochs) %.1f ms = %.1f M interactions/s  [%s]" % (1e3 * dt, data.nnz * 3 / dt / 1e6, " ".join(k + "=" + v for k, v in os.environ.items() if k.startswith("LIGHTFM_AMD_HOT"))), flush=True)
PY
  done
  timeout 2400 python3 tools/hot_gate_sweep.py 8 "off:HOT_SLICES=0" "shipped:X=1" 2>&1 | tail -10
  ;;
r6s)
  # the narrow-model tile kernel (two interactions per lane group): tile suites, then c2 at d = 10 / 12 / 16 with it (4 / 3 / 2 workgroups per CU)
  # and without (LIGHTFM_AMD_TILE_PAIRS=0)
  ( time timeout 1800 $PYT tests/test_hip_warp_tile.py tests/test_baseline_shapes.py -m gpu -x -q ) > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt | cut -c1-300
  line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]; ss = d["config"].get("steady_state", {})
    print("  %-30s %8.2f M/s  frac %.3f  launch %.3f ms  in flight %d  ustore %s | steady %8.2f M/s  %s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["avg_launch_ms"], r["interactions_in_flight"], r.get("user_rows_by_plain_stores"), ss.get("value", 0) / 1e6, r["kernel"][:50]))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5 --steps 10 --warmup 3 --config c2"
  for i in 1 2; do for arm in "d10-wide:LIGHTFM_AMD_TILE_PAIRS=0:10" "d10-pairs:LIGHTFM_AMD_TILE_PAIRS=1:10" "d10-pairs-ustore:LIGHTFM_AMD_TILE_PAIRS=1:10:--debug 2048" "d16-wide:LIGHTFM_AMD_TILE_PAIRS=0:16" "d16-pairs:LIGHTFM_AMD_TILE_PAIRS=1:16" "d16-pairs-ustore:LIGHTFM_AMD_TILE_PAIRS=1:16:--debug 2048"; do
    IFS=: read name envs dd extra <<< "$arm"
    env $envs timeout 300 python3 bench.py $S --no-components $dd $extra > $OUT/c2_${name}_$i.json 2> $OUT/c2_${name}_$i.err; line "c2 $name run $i" $OUT/c2_${name}_$i.json
  done; done
  ;;
r6t)
  # experiment: what the six bias-cell publications of an update cost (debug bit 15 = 32768: not published) -- the write-side line-operation hypothesis
  line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]; ss = d["config"].get("steady_state", {})
    print("  %-30s %8.2f M/s  frac %.3f  launch %.3f ms  U %.3f | steady %8.2f M/s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["avg_launch_ms"], r["updates_per_interaction"], ss.get("value", 0) / 1e6))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5 --steps 10 --warmup 3"
  for i in 1 2; do for arm in "c2-d64:c2:64:" "c2-d64-nobias:c2:64:--debug 32768" "c2-d64-pairedbias:c2:64:--debug 65536" "c2-d10:c2:10:" "c2-d10-pairedbias:c2:10:--debug 65536"; do
    IFS=: read name cfg dd extra <<< "$arm"
    LIGHTFM_AMD_TILE_PAIRS=0 timeout 300 python3 bench.py $S --config $cfg --no-components $dd $extra > $OUT/${name}_$i.json 2> $OUT/${name}_$i.err; line "$name run $i" $OUT/${name}_$i.json
  done; done
  ;;
r6u)
  # the narrow-model kernel on by default: its precision gate at the default width, the tile suites, the default-width leg
  ( time timeout 1500 $PYT tests/test_precision_parity.py -m gpu -q -s -k "default_width or ml100k or tiny" ) > $OUT/gates.txt 2>&1; grep -aE "delta|passed|failed|real" $OUT/gates.txt
  ( time timeout 1800 $PYT tests/test_hip_warp_tile.py tests/test_baseline_shapes.py tests/test_lightfm_api.py tests/test_reference_suite.py -m gpu -x -q ) > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt | cut -c1-300
  ;;
r6v)
  # precision@10 at the default width: reference, narrow-model kernel, wide kernel (40 seeds each); the fixed narrow training test
  timeout 1200 python3 tools/narrow_quality.py 40 2>&1 | tail -3
  NARROW_QUALITY_REF=0 LIGHTFM_AMD_TILE_PAIRS=0 timeout 600 python3 tools/narrow_quality.py 40 2>&1 | tail -1
  ( time timeout 900 $PYT tests/test_hip_warp_tile.py -m gpu -x -q -k narrow ) > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt | cut -c1-300
  ;;
r6w)
  # bias cells as (b, bG) pairs in the steady-state tile kernels: exactness suites, then A/B (LIGHTFM_AMD_BIAS_PAIRS=0) on c2 at d = 64 / 10 and the C4 shard
  ( time timeout 1800 $PYT tests/test_hip_warp_tile.py tests/test_baseline_shapes.py tests/test_hbm_shapes.py tests/test_sharded_items.py tests/test_hip_parity.py tests/test_lightfm_api.py -m gpu -x -q ) > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt | cut -c1-300
  line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]; ss = d["config"].get("steady_state", {})
    print("  %-26s %8.2f M/s  frac %.3f  launch %.3f ms  U %.3f | steady %8.2f M/s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["avg_launch_ms"], r["updates_per_interaction"], ss.get("value", 0) / 1e6))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5 --steps 15 --warmup 5"
  for i in 1 2; do for arm in "c2-d64-pairs:c2:64:1" "c2-d64-separate:c2:64:0" "c2-d10-pairs:c2:10:1" "c2-d10-separate:c2:10:0" "c4shard-pairs:c4shard:64:1" "c4shard-separate:c4shard:64:0"; do
    IFS=: read name cfg dd pairs <<< "$arm"
    LIGHTFM_AMD_BIAS_PAIRS=$pairs timeout 300 python3 bench.py $S --config $cfg --no-components $dd > $OUT/${name}_$i.json 2> $OUT/${name}_$i.err; line "$name run $i" $OUT/${name}_$i.json
  done; done
  ;;
r6x)
  # the narrow-model kernel with W and G of a row in one line (row pairs): tile suites, then c2 at d = 10 / 16 with / without (LIGHTFM_AMD_ROW_PAIRS=0)
  ( time timeout 1800 $PYT tests/test_hip_warp_tile.py tests/test_baseline_shapes.py tests/test_lightfm_api.py -m gpu -x -q ) > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt | cut -c1-300
  line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]; ss = d["config"].get("steady_state", {})
    print("  %-26s %8.2f M/s  frac %.3f  launch %.3f ms  U %.3f  ustore %s | steady %8.2f M/s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["avg_launch_ms"], r["updates_per_interaction"], r.get("user_rows_by_plain_stores"), ss.get("value", 0) / 1e6))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5 --steps 15 --warmup 5 --config c2"
  for i in 1 2; do for arm in "d10-rowpairs:10:1:" "d10-separate:10:0:" "d10-rowpairs-ustore:10:1:--debug 2048" "d16-rowpairs:16:1:" "d16-separate:16:0:"; do
    IFS=: read name dd rp extra <<< "$arm"
    LIGHTFM_AMD_ROW_PAIRS=$rp timeout 300 python3 bench.py $S --no-components $dd $extra > $OUT/${name}_$i.json 2> $OUT/${name}_$i.err; line "c2 $name run $i" $OUT/${name}_$i.json
  done; done
  ;;
r6y)
  # the narrow-model kernel with the bias cells in the row's line as well (LIGHTFM_AMD_ROW_PAIRS=2, d <= 12): tile suites, then c2 at d = 10 / 12 per layout
  ( time timeout 1800 $PYT tests/test_hip_warp_tile.py tests/test_baseline_shapes.py tests/test_lightfm_api.py -m gpu -x -q ) > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt | cut -c1-300
  line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]; ss = d["config"].get("steady_state", {})
    print("  %-26s %8.2f M/s  frac %.3f  launch %.3f ms  U %.3f  ustore %s | steady %8.2f M/s  %s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["avg_launch_ms"], r["updates_per_interaction"], r.get("user_rows_by_plain_stores"), ss.get("value", 0) / 1e6, r.get("kernel")))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 1.5 --steps 15 --warmup 5 --config c2"
  for i in 1 2; do for arm in "d10-rows+biases:10:2:" "d10-rows:10:1:" "d10-separate:10:0:" "d10-rows+biases-ustore:10:2:--debug 2048" "d12-rows+biases:12:2:" "d12-rows:12:1:" "d4-rows+biases:4:2:"; do
    IFS=: read name dd rp extra <<< "$arm"
    LIGHTFM_AMD_ROW_PAIRS=$rp timeout 300 python3 bench.py $S --no-components $dd $extra > $OUT/${name}_$i.json 2> $OUT/${name}_$i.err; line "c2 $name run $i" $OUT/${name}_$i.json
  done; done
  ;;
r6yq)
  # precision@10 at the default width with the bias cells in the row's line (shipped) against the reference and the separate tables; the logistic gate
  timeout 1200 python3 tools/narrow_quality.py 40 2>&1 | tail -3
  NARROW_QUALITY_REF=0 LIGHTFM_AMD_ROW_PAIRS=0 timeout 600 python3 tools/narrow_quality.py 40 2>&1 | tail -1
  ( time timeout 900 $PYT tests/test_precision_parity.py -m gpu -x -q -s -k logistic ) > $OUT/tests.txt 2>&1; grep -a "delta\|passed\|failed" $OUT/tests.txt | cut -c1-300
  ;;
r6la)
  # what the row-stream kernels' atomic LINE operations cost (timing experiments, WRONG results): debug bit 15 = no bias publication,
  # bit 16 = no accumulator rows published -- C5 shard at --scale 0.25 and C3
  line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]; ss = d["config"].get("steady_state", {})
    print("  %-26s %8.2f M/s  frac %.3f  launch %.3f ms  U %.3f | steady %8.2f M/s  %s" % (sys.argv[1], d["value"] / 1e6, r["frac"], r["avg_launch_ms"], r["updates_per_interaction"], ss.get("value", 0) / 1e6, r.get("kernel")))
except Exception as e:
    print("  %s: no result: %r" % (sys.argv[1], e))
PY
  }
  S5="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 0 --steps 2 --warmup 1 --config c5shard --scale 0.25"
  S3="--no-cpu-baseline --no-quality --no-fit --no-extra --fits 1 --steady-seconds 0 --steps 4 --warmup 2 --config c3"
  for i in 1 2; do for dbg in 0 32768 65536 98304; do
    timeout 400 python3 bench.py $S5 --debug $dbg > $OUT/c5_${dbg}_$i.json 2> $OUT/c5_${dbg}_$i.err; line "c5shard debug $dbg run $i" $OUT/c5_${dbg}_$i.json
  done; done
  for dbg in 0 32768 65536 98304; do
    timeout 400 python3 bench.py $S3 --debug $dbg > $OUT/c3_${dbg}.json 2> $OUT/c3_${dbg}.err; line "c3 debug $dbg" $OUT/c3_${dbg}.json
  done
  ;;
r6pa)
  # predict_ranks: the matrix products on the bf16 pipe (timing experiment x4: two-way split operands in the same registers, WRONG ranks)
  # against the fp32 products without exact re-checks (x9) and the shipped kernel; tools/build_variants.sh x4 -DLFM_R3X=4 / x9 -DLFM_R3X=9
  for v in _lib $*; do
    [ -f $R/lightfm_amd/$v/liblfm_hip.so ] || continue
    LIGHTFM_AMD_LIB=$R/lightfm_amd/$v/liblfm_hip.so RANKS_TIMING_MODES=4,3 timeout 200 python tools/ranks_timing.py > $OUT/t$v.txt 2>&1
    echo "$v: $(grep -a "mode [34]" $OUT/t$v.txt | sed -n "3p;6p" | sed 's/.*wall/wall/' | cut -c1-150)"
  done
  ;;
r6z)
  # the driver's sequence on the final tree: GPU suite, smoke, default bench
  ( time timeout 2400 $PYT tests -m gpu -x -q ) > $OUT/suite.txt 2>&1; tail -3 $OUT/suite.txt | cut -c1-300
  timeout 300 python3 __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
  ( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; echo "bench bytes $(wc -c < $OUT/bench.json)"
  grep -a "failed\|Traceback" $OUT/bench.err | head -5
  python3 - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("c2 %.1f M/s frac %.3f  traffic %s" % (d["value"] / 1e6, d["roofline"]["frac"], d["roofline"].get("traffic_over_algorithmic")))
for k, v in d["config"].get("legs", {}).items():
    try: print("  %-10s %s  frac %s" % (k, v.get("value"), (v.get("roofline") or {}).get("frac")))
    except Exception as e: print(k, e)
print("  fit", (d["config"].get("end_to_end_fit") or {}).get("value"), "quality", d["config"].get("quality"))
PY
  ;;
*)
  echo "unknown step $STEP"; exit 2;;
esac
