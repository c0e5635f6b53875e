#!/bin/bash
# Round 2, GPU visit 8: catch the silent abort with the runtime's error message; ranks kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02h
mkdir -p $OUT
cd $R
AMD_LOG_LEVEL=1 LIGHTFM_AMD_REG_SYNC=0 timeout -k 5 100 python tools/stress_launches.py warp 80 > $OUT/stress_warp_nosync.log 2>&1; echo "stress warp nosync exit $?"; tail -4 $OUT/stress_warp_nosync.log | cut -c1-300
AMD_LOG_LEVEL=1 timeout -k 5 100 python tools/stress_launches.py warp-kos 80 0.0 > $OUT/stress_kos.log 2>&1; echo "stress kos exit $?"; tail -4 $OUT/stress_kos.log | cut -c1-300
timeout -k 5 300 python -m pytest tests/test_evaluation_gpu.py tests/test_hip_parity.py -q -x -k "ranks or mfma or evaluation or precision_recall or auc" -p no:cacheprovider > $OUT/pytest_ranks.log 2>&1; echo "ranks tests exit $?"; tail -8 $OUT/pytest_ranks.log | cut -c1-300
timeout -k 5 200 python tools/ranks_timing.py > $OUT/ranks_v2.txt 2>&1; tail -2 $OUT/ranks_v2.txt
LIGHTFM_AMD_RANKS_MFMA=1 timeout -k 5 200 python tools/ranks_timing.py > $OUT/ranks_v1.txt 2>&1; tail -1 $OUT/ranks_v1.txt
