#!/bin/bash
# stage timeline of the fast kernel (ablation build) at 1 and 7 waves per CU, full kernel and without memory instructions
set -o pipefail
OUT=gpurun_out/r02_timeline
mkdir -p $OUT; : > $OUT/timeline.log
ABL=$PWD/l3ster_amd/lib/libl3k_ablation.so
for W in ${WAVES:-1 7}; do
 for F in ${FLAGS:-0 3}; do
  echo "== waves/CU $W flags $F" >> $OUT/timeline.log
  L3K_DEBUG_FLAGS=$F L3K_FAST_WAVES_PER_CU=$W L3K_STAMPS=1 L3K_LIB=$ABL timeout -k 10 200 python tools/kbench.py --order ${ORDER:-6} --ne ${NE:-32} --child 2>&1 | grep -v "amdgpu.ids\|XCD\|lifetimes" >> $OUT/timeline.log || exit 1
 done
done
cat $OUT/timeline.log
