#!/bin/bash
# Round 2, GPU visit 11: two-stage candidate fetch with the LDS-DMA kernel, the multi-GPU emulation with it,
# kernel-trace + PMC profiles of c2 / c3 / c4shard, kernel trace of c5shard.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02k
mkdir -p $OUT
cd $R
Q="--no-cpu-baseline --no-quality --no-fit"
run() { cfg=$1; tag=$2; shift; shift; timeout -k 5 300 env "$@" python bench.py --config $cfg $Q $EXTRA > $OUT/${cfg}_$tag.json 2> $OUT/${cfg}_$tag.err; python - <<PY
import json
try:
    d = json.load(open("$OUT/${cfg}_$tag.json")); r = d["roofline"]
    print("%s %-14s %8.1f M/s frac %.3f in_flight %5d launch %6.2f ms eps %d k/step %.2f S %.2f U %.2f %s" % ("$cfg", "$tag", d["value"]/1e6, r["frac"], r["interactions_in_flight"], r["avg_launch_ms"], d["config"]["epochs_per_step"], r["kernel_time_fraction_of_step"], r["draws_per_interaction"], r["updates_per_interaction"], r.get("phase_cycles_per_pass") or r.get("phase_cycles_per_interaction") or ""))
except Exception as e:
    print("$cfg $tag FAILED", e); import subprocess; print(subprocess.run("tail -3 $OUT/${cfg}_$tag.err", shell=True, capture_output=True, text=True).stdout)
PY
}
E8="--steps 3 --warmup 1 --epochs-per-step 8"
EXTRA="$E8" run c2 default A=1
EXTRA="$E8 --first-batch 5" run c2 fb5 A=1
EXTRA="$E8 --first-batch 7" run c2 fb7 A=1
EMU_SEEDS=1,2,3 EMU_EPOCHS=5 timeout -k 5 600 python tools/multi_gpu_emulation.py 1:sum:4:16384:0 2:adagrad:4:16384:0 4:adagrad:4:16384:0 8:adagrad:4:16384:0 8:sum:4:16384:0 8:mean:4:16384:0 > $OUT/emulation.txt 2>&1
grep "^K=" $OUT/emulation.txt
bash tools/profile2.sh r02k_c2 --config c2
bash tools/profile2.sh r02k_c4shard --config c4shard
bash tools/profile2.sh r02k_c3 --config c3
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02k_c5shard/trace -o trace -- python $R/bench.py --config c5shard --scale 0.25 $Q --steps 2 --warmup 1 --epochs-per-step 1 > $R/gpurun_out/prof_r02k_c5shard/bench_trace.json 2> $R/gpurun_out/prof_r02k_c5shard/bench_trace.err
cd $R && python tools/prof_summary.py r02k_c5shard 2>&1 | head -4 | cut -c1-300
find $R/gpurun_out -size +20M -delete
