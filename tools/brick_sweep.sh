#!/bin/bash
# Mesh-traversal experiment behind the default brick edges (l3ster_amd/csrc/host/cube_mesh.cpp:brickEdge): kernel time vs
# brick edge for orders 6 and 4 on ne^3 elements.  Usage: tools/brick_sweep.sh [ne=64] [edges...]
ne=${1:-64}; shift
edges=${@:-1 2 4 5 6 7 8 9 10 12 16 20 24 32}
for b in $edges; do
  echo "== L3K_MESH_BRICK=$b ne=$ne"
  # (round 4: the brick edge is a compile-time constant; an experiment build per edge, selected with L3K_LIB)
  L3K_VARIANT=brick$b L3K_EXTRA_FLAGS="-DL3K_MESH_BRICK=$b" python -m l3ster_amd.build > /dev/null || exit 1
  L3K_LIB=l3ster_amd/lib/libl3k_brick$b.so timeout -k 10 200 python tools/kbench.py --order 6 --ne $ne --steps 5 --flags 0 --child || exit 1
  L3K_LIB=l3ster_amd/lib/libl3k_brick$b.so timeout -k 10 200 python tools/kbench.py --order 4 --ne $ne --steps 5 --flags 0 --child || exit 1
done
