#!/bin/bash
for ne in 32 48; do
  for b in 4 6 7; do
    echo "== p6 ne=$ne brick=$b"
    L3K_MESH_BRICK=$b timeout -k 10 200 python tools/kbench.py --order 6 --ne $ne --flags 0 --child || exit 1
  done
  for b in 4 12 11; do
    echo "== p4 ne=$ne brick=$b"
    L3K_MESH_BRICK=$b timeout -k 10 200 python tools/kbench.py --order 4 --ne $ne --flags 0 --child || exit 1
  done
done
for b in 5 7 9; do
  echo "== p6 ne=64 brick=$b"
  L3K_MESH_BRICK=$b timeout -k 10 200 python tools/kbench.py --order 6 --ne 64 --steps 5 --flags 0 --child || exit 1
done
for b in 10 11 13 14; do
  echo "== p4 ne=64 brick=$b"
  L3K_MESH_BRICK=$b timeout -k 10 200 python tools/kbench.py --order 4 --ne 64 --steps 5 --flags 0 --child || exit 1
done
