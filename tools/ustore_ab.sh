#!/bin/bash
# tools/ustore_ab.sh -- identity BPR / logistic / WARP at the ML-20M shape, user rows by plain stores (the rule) against atomics (LIGHTFM_AMD_DEBUG=4096)
cd ${GRAFT_REPO_ROOT:-/root/repo}; OUT=gpurun_out/r6ab; mkdir -p $OUT
timeout 200 python3 tools/width_sweep.py bpr,logistic,warp 48,64,128 > $OUT/rule.txt 2>&1 &
LIGHTFM_AMD_DEBUG=4096 timeout 200 python3 tools/width_sweep.py bpr,logistic,warp 48,64,128 > $OUT/atomics_concurrent.txt 2>&1
wait
# (the two ran side by side to warm the box; now one after the other)
timeout 200 python3 tools/width_sweep.py bpr,logistic,warp 48,64,128 > $OUT/rule.txt 2>&1
LIGHTFM_AMD_DEBUG=4096 timeout 200 python3 tools/width_sweep.py bpr,logistic,warp 48,64,128 > $OUT/atomics.txt 2>&1
echo "--- rule"; cut -c1-230 $OUT/rule.txt; echo "--- atomics"; cut -c1-230 $OUT/atomics.txt
