#!/bin/bash
# A/B on one box, order 4 (two elements per wave): the shared face of an x-neighbour pair summed in LDS and scattered once (libl3k.so) vs libl3k_prev.so
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in "" _prev; do
    echo -n "lib$v order 4: "; L3K_LIB=$PWD/l3ster_amd/lib/libl3k$v.so python tools/kbench.py --order 4 --ne 64 --child 2>&1 | grep flags
  done
done
echo -n "lib order 6: "; python tools/kbench.py --order 6 --ne 64 --child 2>&1 | grep flags
