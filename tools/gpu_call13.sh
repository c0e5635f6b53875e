#!/bin/bash
# Round 2, GPU visit 13 (the round's last seconds): the suite without the precision-parity file, RCCL test in its own process.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02m
mkdir -p $OUT
cd $R
timeout -k 5 70 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_precision_parity.py > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log | cut -c1-300
