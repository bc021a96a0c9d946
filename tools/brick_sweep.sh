#!/bin/bash
# element-order experiment: brick size of the mesh traversal vs kernel time (order 6 and 4, 64^3)
for b in 1 2 4 8 16; do
  echo "== L3K_MESH_BRICK=$b"
  L3K_MESH_BRICK=$b timeout -k 10 200 python tools/kbench.py --order 6 --ne 64 --steps 5 --flags 0 --child || exit 1
  L3K_MESH_BRICK=$b timeout -k 10 200 python tools/kbench.py --order 4 --ne 64 --steps 5 --flags 0 --child || exit 1
done
