#!/bin/bash
for b in 8 16 32 64; do
  for x in 0 1; do
    echo "== L3K_MESH_BRICK=$b xcd=$x"
    if [ $x = 1 ]; then export L3K_FAST_XCD=1; unset L3K_FAST_NO_XCD; else unset L3K_FAST_XCD; export L3K_FAST_NO_XCD=1; fi
    L3K_MESH_BRICK=$b timeout -k 10 200 python tools/kbench.py --order 4 --ne 64 --steps 5 --flags 0 --child || exit 1
    L3K_MESH_BRICK=$b timeout -k 10 200 python tools/kbench.py --order 6 --ne 64 --steps 5 --flags 0 --child || exit 1
  done
done
