#!/bin/bash
# tools/nq2.sh -- attribution runs behind profiles/r06_narrow_quality20m.txt (one box): the plain-store user rows at full size, d = 32 and d = 10
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r6nq2
# d = 10 against the first run's reference numbers (same seeds): the narrow kernel / the wide kernel with the user rows FORCED onto atomics
NQ_REF=0.08849,0.00059 NQ_ARMS='[["2 workgroups, atomics", {"LIGHTFM_AMD_NARROW_BLOCKS": "2", "LIGHTFM_AMD_DEBUG": "4096"}], ["wide kernel, atomics", {"LIGHTFM_AMD_TILE_PAIRS": "0", "LIGHTFM_AMD_DEBUG": "4096"}], ["shipped, plain stores", {"LIGHTFM_AMD_DEBUG": "2048"}]]' \
  timeout 700 python tools/narrow_quality20m.py 3 16 10 > gpurun_out/r6nq2/d10.txt 2>&1 &
# d = 32: the shipped plan (wide tile kernel, plain-store user rows) against atomics, with the reference
NQ_ARMS='[["shipped", {}], ["user rows by atomics", {"LIGHTFM_AMD_DEBUG": "4096"}]]' \
  timeout 700 python tools/narrow_quality20m.py 3 16 32 > gpurun_out/r6nq2/d32.txt 2>&1
wait
cat gpurun_out/r6nq2/d10.txt gpurun_out/r6nq2/d32.txt
