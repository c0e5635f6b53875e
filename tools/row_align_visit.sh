#!/bin/bash
# tools/row_align_visit.sh -- rows of 17..63 floats padded to 32 / 48 / 64 (csrc/session.hip) against multiples of 4: rates at the ML-20M shape, then the GPU suite
cd ${GRAFT_REPO_ROOT:-/root/repo}; OUT=gpurun_out/r6ra; mkdir -p $OUT
LIGHTFM_AMD_ROW_ALIGN=0 timeout 100 python3 tools/width_sweep.py warp,bpr,logistic 20,40,56 > $OUT/multiples_of_4.txt 2>&1
timeout 100 python3 tools/width_sweep.py warp,bpr,logistic 20,40,56 > $OUT/aligned.txt 2>&1
echo "--- multiples of 4"; cut -c1-200 $OUT/multiples_of_4.txt; echo "--- aligned"; cut -c1-200 $OUT/aligned.txt
( time timeout 300 python3 -m pytest -p no:cacheprovider tests -m gpu -x -q --deselect tests/test_precision_parity.py ) > $OUT/suite.txt 2>&1
grep -a "passed\|failed\|Error\|assert" $OUT/suite.txt | tail -8 | cut -c1-300
