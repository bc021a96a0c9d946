"""Makes tests/golden/gmsh_cube_hexes.npz from the reference's own test mesh tests/data/gmsh_ascii4_cube.msh (a data
file of the reference's test suite; run in the build container, where /root/reference exists):
vertex coordinates + hex connectivity in the reference's local vertex order v = i + 2j + 4k, and the node count the
reference's test expects after convertMeshToOrder<2> (tests/MeshTests.cpp:249-255: 44745).

gmsh 4.1 ASCII: $Nodes blocks (entity dim, tag, parametric, n) then n tags then n coordinate lines; $Elements blocks
(entity dim, tag, element type, n) then n lines "tag v0 ... ".  Element type 5 = 8-node hexahedron, gmsh corner order
= bottom face counter-clockwise then top face: (0,0,0),(1,0,0),(1,1,0),(0,1,0),(0,0,1),... -> lexicographic order is
gmsh[0,1,3,2,4,5,7,6] (mesh/ReadMesh.hpp reorders the same way).
"""
import os
import sys

import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/tests/data/gmsh_ascii4_cube.msh"
lines = open(src).read().split("\n")
pos = lines.index("$Nodes") + 1
n_blocks, n_nodes = (int(v) for v in lines[pos].split()[:2])
pos += 1
tags, xyz = [], []
for _ in range(n_blocks):
    nb = int(lines[pos].split()[3])
    pos += 1
    tags += [int(lines[pos + i]) for i in range(nb)]
    xyz += [[float(v) for v in lines[pos + nb + i].split()] for i in range(nb)]
    pos += 2 * nb
assert len(tags) == n_nodes
tag_to_idx = {t: i for i, t in enumerate(tags)}
pos = lines.index("$Elements") + 1
n_blocks = int(lines[pos].split()[0])
pos += 1
hexes = []
for _ in range(n_blocks):
    _, _, etype, nb = (int(v) for v in lines[pos].split())
    pos += 1
    if etype == 5:
        for i in range(nb):
            v = [tag_to_idx[int(t)] for t in lines[pos + i].split()[1:]]
            hexes.append([v[0], v[1], v[3], v[2], v[4], v[5], v[7], v[6]])
    pos += nb
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gmsh_cube_hexes.npz")
np.savez_compressed(out, verts=np.array(xyz), conn=np.array(hexes, dtype=np.uint32), order2_node_count=np.int64(44745))
print(out, len(xyz), "vertices", len(hexes), "hexes")
