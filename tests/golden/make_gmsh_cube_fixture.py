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

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from l3ster_amd.gmsh import read_hexes  # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/tests/data/gmsh_ascii4_cube.msh"
verts, conn, _ = read_hexes(src)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gmsh_cube_hexes.npz")
np.savez_compressed(out, verts=verts, conn=conn, order2_node_count=np.int64(44745))
print(out, len(verts), "vertices", len(conn), "hexes")
