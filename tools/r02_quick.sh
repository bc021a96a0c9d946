#!/bin/bash
# quick GPU check of a kernel change: parity tests of the apply, then kbench at order 6 and 4 (64^3)
set -o pipefail
OUT=gpurun_out/r02_quick
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_apply.py -x -q -m gpu > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
for o in 6 4; do
  timeout -k 10 300 python tools/kbench.py --order $o --ne 64 --child 2>&1 | grep flags | tee -a $OUT/kbench.log
done
