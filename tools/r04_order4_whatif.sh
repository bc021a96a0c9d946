#!/bin/bash
# Order-4 apply (BASELINE configs[1]): what the 2x2x2 element-block idea could save at most.  The block kernel would sum the shared
# faces of 8 elements in LDS: 386 instead of 784 shell-node updates per 8 elements = -51 % atomic adds.  Ablation flag 128 drops every
# other shell round of the scatter (-50 % atomics AND their LDS reads / address arithmetic, which the block kernel would NOT save;
# wrong results, valid timing): an upper bound of the gain.  Beside it: no scatter memory instructions at all (1), plain stores (16).
export TMPDIR=/tmp
for o in 4 6; do
  python tools/kbench.py --order $o --ne 64 --steps 10 --flags 0,128,16,1,2,3 || exit 1
done
