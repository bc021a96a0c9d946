#!/bin/bash
# A/B on one box: Dirichlet dofs zeroed at staging, one scatter path for all elements (libl3k.so) vs the round-2 form with a rolled
# scatter path for elements that touch a Dirichlet dof (libl3k_flagged.so); orders 6 and 4 at 64^3, alternating, three times
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in "" _flagged; do
    for o in 6 4; do
      echo -n "lib$v order $o: "; L3K_LIB=$PWD/l3ster_amd/lib/libl3k$v.so python tools/kbench.py --order $o --ne 64 --child 2>&1 | grep flags
    done
  done
done
