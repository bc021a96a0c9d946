"""Independent numpy/mpmath restatement of the L3STER element-local formulas.  TEST INFRASTRUCTURE, NOT PRODUCT.

Purpose: a second, structurally different statement of the same mathematics (dense B_q matrices built with einsum,
tables from mpmath at 50 digits) used to (a) cross-check oracle/oracle.cpp and (b) generate the golden fixtures under
tests/golden/ (oracle/make_golden.py).  It follows the reference's definitions, cited as file:line under
/root/reference:

  K_e = sum_q w_q detJ_q B_q^T B_q,  F_e = sum_q w_q detJ_q B_q^T f_q        algsys/AssembleLocalSystem.hpp:131-166
  B_q[:, b*U+u] = A0[:,u] phi_b + sum_d A_d[:,u] d_d phi_b                    algsys/AssembleLocalSystem.hpp:131-142
  y_e = K_e x_e                                                               algsys/EvaluateLocalOperator.hpp:130-146
  diag = diag(K_e), rhs = F_e - K_e[:, D] g_D                                 algsys/EvaluateLocalOperator.hpp:172-208
  phi_b tensor Lagrange on GLL nodes, b = ix + n(iy + n iz)                   basisfun/ReferenceBasisFunction.hpp:74-153
  J[d][s] = d x_s / d xi_d from the 2^dim vertices, grad phi = J^{-1} grad_ref phi   mapping/JacobiMat.hpp:15-45,
                                                                               mapping/ComputePhysBasisDer.hpp:9-15
"""
import numpy as np

try:
    import mpmath as mp
    mp.mp.dps = 50
except ImportError:  # pragma: no cover
    mp = None


# ---------------------------------------------------------------- tables (mpmath, 50 digits)
def _legendre_roots(n, derivative=False):
    """Roots of P_n (or of P_n') ascending: double-precision guesses from numpy's Legendre series (companion-matrix
    eigenvalues, the reference's own route, math/Polynomial.hpp:98-122) polished to 50 digits by mp.findroot."""
    series = np.polynomial.legendre.Legendre.basis(n)
    guesses = np.sort((series.deriv() if derivative else series).roots().real)
    if derivative:
        f = lambda x: mp.diff(lambda t: mp.legendre(n, t), x)
    else:
        f = lambda x: mp.legendre(n, x)
    roots = sorted(mp.findroot(f, mp.mpf(float(g))) for g in guesses)
    assert all(abs(float(r) - g) < 1e-10 for r, g in zip(roots, guesses))
    return roots


def gll_nodes_mp(n):
    if n == 2:
        return [mp.mpf(-1), mp.mpf(1)]
    return [mp.mpf(-1)] + _legendre_roots(n - 1, derivative=True) + [mp.mpf(1)]


def gl_rule_mp(nq):
    xs = _legendre_roots(nq)
    ws = []
    for x in xs:
        dP = mp.diff(lambda t: mp.legendre(nq, t), x)
        ws.append(2 / ((1 - x * x) * dP * dP))
    return xs, ws


def lagrange_mp(nodes, x):
    n = len(nodes)
    vals, ders = [], []
    for b in range(n):
        den = mp.mpf(1)
        for j in range(n):
            if j != b:
                den *= nodes[b] - nodes[j]
        v = mp.mpf(1)
        for j in range(n):
            if j != b:
                v *= x - nodes[j]
        d = mp.mpf(0)
        for k in range(n):
            if k == b:
                continue
            pr = mp.mpf(1)
            for j in range(n):
                if j != b and j != k:
                    pr *= x - nodes[j]
            d += pr
        vals.append(v / den)
        ders.append(d / den)
    return vals, ders


def tables(p, nq):
    """gll[n], qx[nq], qw[nq], I[n,nq], D[n,nq] as float64 (rounded from 50-digit values)."""
    gll = gll_nodes_mp(p + 1)
    qx, qw = gl_rule_mp(nq)
    I = np.zeros((p + 1, nq))
    D = np.zeros((p + 1, nq))
    for q, x in enumerate(qx):
        v, d = lagrange_mp(gll, x)
        I[:, q] = [float(t) for t in v]
        D[:, q] = [float(t) for t in d]
    f = lambda a: np.array([float(t) for t in a])
    return f(gll), f(qx), f(qw), I, D


# ---------------------------------------------------------------- kernels (restated; ids as oracle.h)
def kernel_params(kid):
    return {0: (3, 7, 4, 0), 1: (3, 7, 4, 1), 2: (2, 4, 3, 0), 3: (2, 4, 3, 1), 4: (3, 7, 4, 3),
            10: (3, 7, 4, 0), 11: (3, 1, 1, 3), 12: (3, 4, 3, 0), 13: (3, 8, 7, 7)}[kid]


def eval_kernel(kid, fv, fd, R, kparams=None, point=(0.0, 0.0, 0.0), time=0.0):
    """Returns A [D+1,E,U], rhs [E,R].  tests/Kernels.hpp:5-118, benchmarks/Diffusion3D.hpp:51-79; ids 10-12: the synthetic
    point / time reading kernels of oracle.h; 13: benchmarks/Kernels.hpp:3-65 (NS3D)."""
    dim, E, U, F = kernel_params(kid)
    A = np.zeros((dim + 1, E, U))
    rhs = np.zeros((E, R))
    x, y, z = point
    if kid == 10:
        k0, s0 = (1.0, 1.0) if kparams is None else kparams[:2]
        lam = k0 * (1.0 + 0.3 * np.sin(x + 2 * y + 3 * z + time))
        c = -(1.0 + 0.2 * np.cos(z - time))
        rhs[0, 0] = s0 * (1.0 + x * y - 0.5 * z * time)
        A[1, 0, 1] = A[2, 0, 2] = A[3, 0, 3] = -lam
        for d in (1, 2, 3):
            A[0, d, d] = c
            A[d, d, 0] = 1.0
        A[2, 4, 3], A[3, 4, 2] = 1.0, -1.0
        A[1, 5, 3], A[3, 5, 1] = -1.0, 1.0
        A[1, 6, 2], A[2, 6, 1] = 1.0, -1.0
    elif kid == 11:
        dt = 0.02 if kparams is None else kparams[0]
        ys, zs = 2 * y - 1, 2 * z - 1
        A[0, 0, 0] = 11.0 / 6.0
        A[1, 0, 0] = (1 - ys * ys) * (1 - 0.5 * zs * zs) * dt
        A[2, 0, 0] = 0.25 * x * dt
        A[3, 0, 0] = -0.125 * dt
        rhs[0, 0] = 3.0 * fv[0] - 1.5 * fv[1] + fv[2] / 3.0
    elif kid == 12:
        f = 1.0 if kparams is None else kparams[0]
        a = 1.0 + 0.5 * x * z
        A[1, 0, 0] = A[2, 0, 1] = A[3, 0, 2] = a
        rhs[0, 0] = f
        A[2, 1, 2], A[3, 1, 1] = 1.0, -1.0
        A[3, 2, 0], A[1, 2, 2] = 1.0, -1.0
        A[1, 3, 1], A[2, 3, 0] = 1.0, -1.0
        A[0, 1, 0] = 0.1 * y
        rhs[1, 0] = 0.5
    elif kid == 13:
        ri = 1e-3
        u, v, w = fv[0], fv[1], fv[2]
        (ux, vx, wx), (uy, vy, wy), (uz, vz, wz) = fd[0][:3], fd[1][:3], fd[2][:3]
        A[0, 0, :3], A[0, 1, :3], A[0, 2, :3] = [ux, uy, uz], [vx, vy, vz], [wx, wy, wz]
        A[0, 3, 4] = A[0, 4, 5] = A[0, 5, 6] = 1.0
        for (d, vel) in ((1, u), (2, v), (3, w)):
            A[d, 0, 0] = A[d, 1, 1] = A[d, 2, 2] = vel
            A[d, 0, 3] = 1.0
        A[1, 1, 6], A[1, 2, 5], A[1, 4, 2], A[1, 5, 1], A[1, 6, 0], A[1, 7, 4] = -ri, ri, -1.0, 1.0, 1.0, 1.0
        A[2, 0, 6], A[2, 2, 4], A[2, 3, 2], A[2, 5, 0], A[2, 6, 1], A[2, 7, 5] = ri, -ri, 1.0, -1.0, 1.0, 1.0
        A[3, 0, 5], A[3, 1, 4], A[3, 3, 1], A[3, 4, 0], A[3, 6, 2], A[3, 7, 6] = -ri, ri, -1.0, 1.0, 1.0, 1.0
        # (benchmarks/Kernels.hpp:27,38,49: the pressure gradient enters the three momentum rows: A1(0,3), A2(0,3), A3(0,3))
        rhs[0, 0] = u * ux + v * uy + w * uz
        rhs[1, 0] = u * vx + v * vy + w * vz
        rhs[2, 0] = u * wx + v * wy + w * wz
    elif kid in (0, 1, 4):
        if kid == 0:
            k, s = (1.0, 1.0) if kparams is None else kparams[:2]
            lam = k
            rhs[0, 0] = s
        elif kid == 1:
            lam = fv[0]
            A[0, 0, 1:4] = [-fd[0][0], -fd[1][0], -fd[2][0]]
        else:
            k, sigma, s = (1.0, 1.0, 1.0) if kparams is None else kparams[:3]
            lam = k
            A[0, 0, 0] = sigma
            A[1, 0, 0], A[2, 0, 0], A[3, 0, 0] = fv[0], fv[1], fv[2]
            rhs[0, 0] = s
        A[1, 0, 1] = A[2, 0, 2] = A[3, 0, 3] = -lam
        for d in (1, 2, 3):
            A[0, d, d] = -1.0
            A[d, d, 0] = 1.0
        A[2, 4, 3], A[3, 4, 2] = 1.0, -1.0
        A[1, 5, 3], A[3, 5, 1] = -1.0, 1.0
        A[1, 6, 2], A[2, 6, 1] = 1.0, -1.0
    else:
        lam = 1.0 if kid == 2 else fv[0]
        if kid == 3:
            A[0, 0, 1], A[0, 0, 2] = -fd[0][0], -fd[1][0]
        A[1, 0, 1] = A[2, 0, 2] = -lam
        A[0, 1, 1] = A[0, 2, 2] = -1.0
        A[1, 1, 0] = A[2, 2, 0] = 1.0
        A[1, 3, 2], A[2, 3, 1] = 1.0, -1.0
    return A, rhs


# ---------------------------------------------------------------- element operators (dense)
def _lin(x):
    return np.array([0.5 * (1 - x), 0.5 * (1 + x)]), np.array([-0.5, 0.5])


def jacobi(dim, verts, pt):
    """J[d, s]"""
    v, d = zip(*[_lin(pt[a]) for a in range(dim)])
    J = np.zeros((dim, dim))
    for vi in range(2 ** dim):
        idx = [(vi >> a) & 1 for a in range(dim)]
        for dd in range(dim):
            sf = 1.0
            for a in range(dim):
                sf *= d[a][idx[a]] if a == dd else v[a][idx[a]]
            J[dd, :] += verts[vi, :dim] * sf
    return J


def phys_point(dim, verts, pt):
    v = [_lin(pt[a])[0] for a in range(dim)]
    out = np.zeros(3)
    for vi in range(2 ** dim):
        sf = 1.0
        for a in range(dim):
            sf *= v[a][(vi >> a) & 1]
        out += verts[vi] * sf
    return out


def element_B(kid, p, nq, R, verts, node_fields=None, kparams=None, time=0.0):
    """Returns B [nqp, E, Nd], w*detJ [nqp], f [nqp, E, R]; QP order x fastest (order is irrelevant to the sums)."""
    dim, E, U, F = kernel_params(kid)
    verts = np.asarray(verts, dtype=np.float64)
    gll, qx, qw, I, D = tables(p, nq)
    n = p + 1
    N = n ** dim
    nqp = nq ** dim
    B = np.zeros((nqp, E, N * U))
    wj = np.zeros(nqp)
    fq = np.zeros((nqp, E, R))
    for qi in range(nqp):
        q = [(qi // nq ** a) % nq for a in range(dim)]
        pt = [qx[q[a]] for a in range(dim)]
        # tensor basis values / reference derivatives, b = ix + n*(iy + n*iz)
        phi = np.ones(N)
        dphi = np.ones((dim, N))
        for b in range(N):
            bi = [(b // n ** a) % n for a in range(dim)]
            for a in range(dim):
                phi[b] *= I[bi[a], q[a]]
                for dd in range(dim):
                    dphi[dd, b] *= D[bi[a], q[a]] if a == dd else I[bi[a], q[a]]
        J = jacobi(dim, verts, pt)
        gphi = np.linalg.solve(J, dphi)  # J^{-1} grad_ref
        fv = np.zeros(F)
        fd = np.zeros((dim, F))
        if F:
            fv = node_fields.T @ phi
            fd = gphi @ node_fields
        A, rhs = eval_kernel(kid, fv, fd, R, kparams, phys_point(dim, verts, pt), time)
        Bq = np.einsum("eu,b->ebu", A[0], phi)
        for dd in range(dim):
            Bq += np.einsum("eu,b->ebu", A[dd + 1], gphi[dd])
        B[qi] = Bq.reshape(E, N * U)
        w = 1.0
        for a in range(dim):
            w *= qw[q[a]]
        wj[qi] = w * np.linalg.det(J)
        fq[qi] = rhs
    return B, wj, fq


def assemble(kid, p, nq, R, verts, node_fields=None, kparams=None, time=0.0):
    B, wj, fq = element_B(kid, p, nq, R, verts, node_fields, kparams, time)
    nqp, E, Nd = B.shape
    Bw = (B * wj[:, None, None]).reshape(nqp * E, Nd)
    K = Bw.T @ B.reshape(nqp * E, Nd)
    Fe = Bw.T @ fq.reshape(nqp * E, R)
    return K, Fe


def node_locations(dim, p, verts):
    gll = tables(p, p + 1)[0]
    n = p + 1
    out = np.zeros((n ** dim, 3))
    for b in range(n ** dim):
        pt = [gll[(b // n ** a) % n] for a in range(dim)]
        out[b] = phys_point(dim, np.asarray(verts, dtype=np.float64), pt)
    return out


def boundary_nodes(dim, p):
    """Local indices of nodes on the element boundary (mesh/ElementTraits.hpp boundary_node_inds), ascending."""
    n = p + 1
    out = []
    for b in range(n ** dim):
        bi = [(b // n ** a) % n for a in range(dim)]
        if any(i in (0, n - 1) for i in bi):
            out.append(b)
    return np.array(out)


# ---- native results file: restatement of post/NativeIO.hpp (writer :15-60, reader LoadedResults :115-146) -------------
def results_file_bytes(fields, comment=""):
    """The whole file for field-major values [n_fields][n_nodes_global]: header text, two size_t, the doubles."""
    f = np.ascontiguousarray(fields, dtype=np.float64)
    header = ("L3STER results file\nv1.0\n// %s\n" % comment.replace("\n", " ")).encode()  # :38-39
    header += np.array([f.shape[0], f.shape[1]], dtype=np.uint64).tobytes()  # util::serialize(size_t) x2, :40-41
    return header + f.tobytes()  # field i at header + 8 * n_nodes * i  (:50-52)


def results_file_parse(data):
    """LoadedResults: skip 3 lines (:120-121), read (n_fields, n_nodes) (:124), values(node, field) at
    8 * (n_nodes * field + node) (:133-139).  Returns (comment, [n_fields][n_nodes])."""
    pos = 0
    lines = []
    for _ in range(3):
        nl = data.index(b"\n", pos)
        lines.append(data[pos:nl])
        pos = nl + 1
    n_fields, n_nodes = (int(v) for v in np.frombuffer(data[pos:pos + 16], dtype=np.uint64))
    vals = np.frombuffer(data[pos + 16:pos + 16 + 8 * n_fields * n_nodes], dtype=np.float64).reshape(n_fields, n_nodes)
    assert lines[0] == b"L3STER results file" and lines[1] == b"v1.0" and lines[2].startswith(b"// ")
    return lines[2][3:].decode(), vals


# ---- native mesh file: restatement of post/NativeIO.hpp:75-108 (writer), :161-182 (header), mesh/MeshUtils.hpp:318-360
# (serializeMesh / deserializeMesh) and util/Serialization.hpp:20-66 (object representation / count + values / members in
# order).  The reference holds no mesh file among its test data: the format is pinned by reading the source only.
_MESH_DIMS = (3, 2, 1)  # Hex, Quad, Line: the order of mesh/ElementType.hpp:11-16 in every domain's element tuple


def mesh_part_bytes(order, domains, nodes_begin, n_owned, boundary_ids):
    """domains: {id: [hex, quad, line]} with each entry None or (nodes [n][(p+1)^d] u64, verts [n][2^d][3] f64, ids [n]
    u64).  An element is its object representation {nodes; vertices; id} (mesh/Element.hpp:29-31), no padding."""
    out = [np.uint64(len(domains)).tobytes()]
    for dom_id in sorted(domains):  # std::map< d_id_t, Domain > (MeshPartition.hpp:49)
        out.append(np.uint16(dom_id).tobytes())  # pair.first, 2 bytes, the pair is serialised member by member
        for dim, el in zip(_MESH_DIMS, domains[dom_id]):
            n = 0 if el is None else len(el[2])
            out.append(np.uint64(n).tobytes())
            for i in range(n):
                out.append(np.asarray(el[0][i], dtype=np.uint64).reshape((order + 1) ** dim).tobytes())
                out.append(np.asarray(el[1][i], dtype=np.float64).reshape(2 ** dim * 3).tobytes())
                out.append(np.uint64(el[2][i]).tobytes())
    out.append(np.uint64(nodes_begin).tobytes())
    out.append(np.uint64(n_owned).tobytes())
    b = np.asarray(boundary_ids, dtype=np.uint16)
    out.append(np.uint64(b.size).tobytes() + b.tobytes())
    return b"".join(out)


def mesh_file_bytes(parts, comment=""):
    """The whole file for the serialised parts (bytes each)."""
    header = ("L3STER mesh file\nv1.0\n// %s\n" % comment.replace("\n", " ")).encode()  # :91-92
    header += np.array([len(parts)] + [len(p) for p in parts], dtype=np.uint64).tobytes()  # :94-96
    return header + b"".join(parts)  # part r at header + sum of the sizes before it (:100-101)


def mesh_file_parse(data, order):
    """extractSavedPartitionInfo + deserializeMesh of every part.  Returns (comment, [(domains, nodes_begin, n_owned,
    boundary_ids)])."""
    pos, lines = 0, []
    for _ in range(3):
        nl = data.index(b"\n", pos)
        lines.append(data[pos:nl])
        pos = nl + 1
    assert lines[0] == b"L3STER mesh file" and lines[1] == b"v1.0" and lines[2].startswith(b"// ")
    u64 = lambda at: int(np.frombuffer(data[at:at + 8], dtype=np.uint64)[0])
    n_parts = u64(pos)
    sizes = [u64(pos + 8 + 8 * i) for i in range(n_parts)]
    pos += 8 + 8 * n_parts
    parts = []
    for sz in sizes:
        end, q = pos + sz, pos
        doms = {}
        n_dom = u64(q)
        q += 8
        for _ in range(n_dom):
            dom_id = int(np.frombuffer(data[q:q + 2], dtype=np.uint16)[0])
            q += 2
            els = []
            for dim in _MESH_DIMS:
                n = u64(q)
                q += 8
                nn, nv = (order + 1) ** dim, 2 ** dim
                rec = np.frombuffer(data[q:q + n * 8 * (nn + 3 * nv + 1)], dtype=np.uint64).reshape(n, nn + 3 * nv + 1)
                q += rec.nbytes
                els.append((rec[:, :nn].copy(), rec[:, nn:nn + 3 * nv].copy().view(np.float64).reshape(n, nv, 3),
                            rec[:, -1].copy()) if n else None)
            doms[dom_id] = els
        nodes_begin, n_owned, nb = u64(q), u64(q + 8), u64(q + 16)
        bnd = np.frombuffer(data[q + 24:q + 24 + 2 * nb], dtype=np.uint16).copy()
        assert q + 24 + 2 * nb == end
        parts.append((doms, nodes_begin, n_owned, bnd))
        pos = end
    return lines[2][3:].decode(), parts


def make_cube_mesh(dist):
    """makeCubeMesh(dist) (mesh/primitives/CubeMesh.hpp:16-138) as mesh-file domains: order-1 hexes of domain 0 and the
    boundary quads of domains 1..6 (back z0, front z1, bottom y0, top y1, left x0, right x1), node ids x fastest, element
    ids in emplacement order.  Returns (domains, n_nodes)."""
    d = np.asarray(dist, dtype=np.float64)
    n = d.size
    e = n - 1
    doms = {k: ([], [], []) for k in range(7)}
    nid = lambda ix, iy, iz: n * n * iz + n * iy + ix
    pt = lambda ix, iy, iz: (d[ix], d[iy], d[iz])
    el = 0

    def emplace(dom, corners):
        nonlocal el
        doms[dom][0].append([nid(*c) for c in corners])
        doms[dom][1].append([pt(*c) for c in corners])
        doms[dom][2].append(el)
        el += 1

    for iz in range(e):
        for iy in range(e):
            for ix in range(e):
                emplace(0, [(ix + a, iy + b, iz + c) for c in (0, 1) for b in (0, 1) for a in (0, 1)])  # :44-61
    for iy in range(e):
        for ix in range(e):
            for dom, z in ((1, 0), (2, e)):  # :70-90: back, front
                emplace(dom, [(ix + a, iy + b, z) for b in (0, 1) for a in (0, 1)])
    for iz in range(e):
        for ix in range(e):
            for dom, y in ((3, 0), (4, e)):  # :95-113: bottom, top
                emplace(dom, [(ix + a, y, iz + c) for c in (0, 1) for a in (0, 1)])
    for iz in range(e):
        for iy in range(e):
            for dom, x in ((5, 0), (6, e)):  # :118-136: left, right
                emplace(dom, [(x, iy + b, iz + c) for c in (0, 1) for b in (0, 1)])
    out = {}
    for k, (nodes, verts, ids) in doms.items():
        el_k = (np.array(nodes, dtype=np.uint64), np.array(verts, dtype=np.float64), np.array(ids, dtype=np.uint64))
        out[k] = [el_k, None, None] if k == 0 else [None, el_k, None]
    return out, n ** 3


# ---- order elevation of an order-1 hex mesh: restatement of the SPECIFICATION in l3ster_amd/csrc/api_mesh.hip -----------
# (the reference's convertMeshToOrder, mesh/ConvertMeshToOrder.hpp:51-104, numbers by traversal order; what it and this
# share -- and what tests/MeshTests.cpp:244-279 checks -- is the topology: one node per vertex, p-1 per edge, (p-1)^2 per
# face, (p-1)^3 per element, ids contiguous)
def _edge_verts(e):
    d, q = e >> 2, e & 3
    c0, c1 = q & 1, q >> 1
    if d == 0:
        va = 2 * c0 + 4 * c1
        return va, va + 1
    if d == 1:
        va = c0 + 4 * c1
        return va, va + 2
    va = c0 + 2 * c1
    return va, va + 4


def _face_vert(f, s, t):
    hi = f & 1
    return [s + 2 * t + 4 * hi, s + 4 * t + 2 * hi, 2 * s + 4 * t + hi][f >> 1]


def elevate_order(conn, n_vertices, p):
    """conn [n_elems][8] (local vertex v = i + 2j + 4k) -> (elem_nodes [n_elems][(p+1)^3], n_nodes, n_noninternal)."""
    conn = np.asarray(conn, dtype=np.int64).reshape(-1, 8)
    ne, n1, m = conn.shape[0], p + 1, p - 1
    ev = np.array([_edge_verts(e) for e in range(12)])
    ekeys = np.sort(conn[:, ev], axis=2).reshape(-1, 2)  # (min, max) per (element, edge)
    fv = np.array([[_face_vert(f, q & 1, q >> 1) for q in range(4)] for f in range(6)])
    fkeys = np.sort(conn[:, fv], axis=2).reshape(-1, 4)
    if ne:
        _, edge_id = np.unique(ekeys, axis=0, return_inverse=True)  # lexicographic order of the unique keys
        _, face_id = np.unique(fkeys, axis=0, return_inverse=True)
        edge_id, face_id = edge_id.reshape(ne, 12), face_id.reshape(ne, 6)
        n_edges, n_faces = int(edge_id.max()) + 1, int(face_id.max()) + 1
    else:
        edge_id, face_id, n_edges, n_faces = np.zeros((0, 12), int), np.zeros((0, 6), int), 0, 0
    edge_base, face_base = n_vertices, n_vertices + n_edges * m
    int_base = face_base + n_faces * m * m
    out = np.empty((ne, n1 ** 3), dtype=np.int64)
    for k in range(n1):
        for j in range(n1):
            for i in range(n1):
                ln = i + n1 * (j + n1 * k)
                bi, bj, bk = i in (0, p), j in (0, p), k in (0, p)
                nb = bi + bj + bk
                if nb == 3:
                    out[:, ln] = conn[:, (1 if i else 0) + 2 * (1 if j else 0) + 4 * (1 if k else 0)]
                elif nb == 2:
                    if not bi:
                        le, t = (1 if j else 0) + 2 * (1 if k else 0), i
                    elif not bj:
                        le, t = 4 + (1 if i else 0) + 2 * (1 if k else 0), j
                    else:
                        le, t = 8 + (1 if i else 0) + 2 * (1 if j else 0), k
                    va, vb = _edge_verts(le)
                    pos = np.where(conn[:, va] < conn[:, vb], t - 1, p - 1 - t)
                    out[:, ln] = edge_base + edge_id[:, le] * m + pos
                elif nb == 1:
                    if bk:
                        f, s, t = (1 if k else 0), i, j
                    elif bj:
                        f, s, t = 2 + (1 if j else 0), i, k
                    else:
                        f, s, t = 4 + (1 if i else 0), j, k
                    corners = conn[:, [_face_vert(f, q & 1, q >> 1) for q in range(4)]]
                    o = np.argmin(corners, axis=1)  # first minimum = the device loop's strict '<'
                    os_, ot = o & 1, o >> 1
                    rows = np.arange(ne)
                    ns = corners[rows, (1 - os_) + 2 * ot]
                    nt = corners[rows, os_ + 2 * (1 - ot)]
                    ds = np.where(os_ == 1, p - s, s)
                    dt = np.where(ot == 1, p - t, t)
                    a = np.where(ns < nt, ds, dt)
                    b = np.where(ns < nt, dt, ds)
                    out[:, ln] = face_base + face_id[:, f] * m * m + (a - 1) + m * (b - 1)
                else:
                    out[:, ln] = int_base + np.arange(ne) * m ** 3 + (i - 1) + m * ((j - 1) + m * (k - 1))
    return out, int_base + ne * m ** 3, int_base
