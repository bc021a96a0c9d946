#!/bin/bash
for b in 6 10 12; do
  echo "== p6 L3K_MESH_BRICK=$b"
  L3K_MESH_BRICK=$b timeout -k 10 200 python tools/kbench.py --order 6 --ne 64 --steps 5 --flags 0 --child || exit 1
done
for b in 12 20 24; do
  echo "== p4 L3K_MESH_BRICK=$b"
  L3K_MESH_BRICK=$b timeout -k 10 200 python tools/kbench.py --order 4 --ne 64 --steps 5 --flags 0 --child || exit 1
done
echo "== p6 ne=32 brick 8 / 4"
L3K_MESH_BRICK=8 timeout -k 10 200 python tools/kbench.py --order 6 --ne 32 --flags 0 --child
L3K_MESH_BRICK=4 timeout -k 10 200 python tools/kbench.py --order 6 --ne 32 --flags 0 --child
echo "== p4 ne=48 brick 16 / 4"
L3K_MESH_BRICK=16 timeout -k 10 200 python tools/kbench.py --order 4 --ne 48 --flags 0 --child
L3K_MESH_BRICK=4 timeout -k 10 200 python tools/kbench.py --order 4 --ne 48 --flags 0 --child
