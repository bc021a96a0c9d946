#!/bin/bash
# round-2 experiment batch 1: FP64 vector/matrix co-execution; stage timeline of the fast kernel by resident waves per CU
set -o pipefail
OUT=gpurun_out/r02_exp1
mkdir -p $OUT
# the probes are built from their sources here (no binaries in the repository)
for t in fp64_coissue fp64_issue lds_write_overlap; do
  hipcc -O3 --offload-arch=gfx950 tools/$t.hip -o tools/$t || exit 1
done
timeout -k 10 120 tools/fp64_coissue > $OUT/fp64_coissue.log 2>&1 || { echo "coissue failed"; tail -5 $OUT/fp64_coissue.log; }
cat $OUT/fp64_coissue.log
ABL=$PWD/l3ster_amd/lib/libl3k_ablation.so
for W in 1 2 4 7; do
  echo "== waves/CU $W, stamps, full kernel" >> $OUT/timeline.log
  L3K_FAST_WAVES_PER_CU=$W L3K_STAMPS=1 L3K_LIB=$ABL timeout -k 10 200 python tools/kbench.py --order 6 --ne 32 --child >> $OUT/timeline.log 2>&1 || exit 1
  echo "== waves/CU $W, stamps, no gather / no scatter memory instructions" >> $OUT/timeline.log
  L3K_DEBUG_FLAGS=3 L3K_FAST_WAVES_PER_CU=$W L3K_STAMPS=1 L3K_LIB=$ABL timeout -k 10 200 python tools/kbench.py --order 6 --ne 32 --child >> $OUT/timeline.log 2>&1 || exit 1
done
echo "== production library, 64^3" >> $OUT/timeline.log
timeout -k 10 300 python tools/kbench.py --order 6 --ne 64 --child >> $OUT/timeline.log 2>&1 || exit 1
timeout -k 10 300 python tools/kbench.py --order 4 --ne 64 --child >> $OUT/timeline.log 2>&1 || exit 1
cat $OUT/timeline.log
