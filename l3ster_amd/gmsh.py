"""Hexahedra of a gmsh 4.1 ASCII mesh file: vertex coordinates and the connectivity in the reference's local vertex order
v = i + 2j + 4k (mesh/ReadMesh.hpp reads gmsh 2.2 / 4.1 files and reorders the same way; only what the device path needs:
$Nodes and the type-5 blocks of $Elements -- lower-dimensional boundary elements are faces of the hexahedra and carry no
nodes of their own)."""
import numpy as np

# gmsh corner order of an 8-node hexahedron: bottom face counter-clockwise, then the top face -> lexicographic
GMSH_HEX_TO_LEX = (0, 1, 3, 2, 4, 5, 7, 6)


def read_hexes(path):
    """Returns (verts float64 [n][3], conn uint32 [n_hexes][8], entity int32 [n_hexes]: the volume entity tag of each hex)."""
    lines = open(path).read().split("\n")
    if "$MeshFormat" not in lines or not lines[lines.index("$MeshFormat") + 1].startswith("4.1 0"):
        raise ValueError("only gmsh 4.1 ASCII files are read")
    pos = lines.index("$Nodes") + 1
    n_blocks, n_nodes = (int(v) for v in lines[pos].split()[:2])
    pos += 1
    tags, xyz = [], []
    for _ in range(n_blocks):
        nb = int(lines[pos].split()[3])
        pos += 1
        tags += [int(lines[pos + i]) for i in range(nb)]
        xyz += [[float(v) for v in lines[pos + nb + i].split()[:3]] for i in range(nb)]
        pos += 2 * nb
    if len(tags) != n_nodes:
        raise ValueError("node count mismatch")
    index_of = {t: i for i, t in enumerate(tags)}
    pos = lines.index("$Elements") + 1
    n_blocks = int(lines[pos].split()[0])
    pos += 1
    conn, entity = [], []
    for _ in range(n_blocks):
        _, ent, etype, nb = (int(v) for v in lines[pos].split())
        pos += 1
        if etype == 5:
            for i in range(nb):
                v = [index_of[int(t)] for t in lines[pos + i].split()[1:9]]
                conn.append([v[g] for g in GMSH_HEX_TO_LEX])
                entity.append(ent)
        pos += nb
    return np.array(xyz, dtype=np.float64).reshape(-1, 3), np.array(conn, dtype=np.uint32).reshape(-1, 8), np.array(entity, dtype=np.int32)
