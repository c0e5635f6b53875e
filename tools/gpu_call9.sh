#!/bin/bash
# Round 2, GPU visit 9: the suite (without the precision-parity file) under the abort-trace shim and
# AMD_LOG_LEVEL=1, predict_ranks timing with (tile, pass) work items.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02i
mkdir -p $OUT
cd $R
timeout -k 5 200 python tools/ranks_timing.py > $OUT/ranks_v2.txt 2>&1; tail -2 $OUT/ranks_v2.txt
AMD_LOG_LEVEL=1 LD_PRELOAD=$R/tools/_bin/libaborttrace.so timeout -k 5 900 python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider --deselect tests/test_precision_parity.py > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
grep -v "^  File\|Extension modules\|Unknown Event Type" $OUT/pytest.log | grep -n -E "abort_trace|\.so|FAILED|ERROR|passed|failed|s call|pytest exit|rocdevice|Fatal" | head -80
