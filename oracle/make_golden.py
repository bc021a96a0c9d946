"""Generates tests/golden/*.npz from the independent numpy/mpmath restatement (oracle/oracle_np.py).

Run from the repo root:  python oracle/make_golden.py
The reference holds no stored golden vectors; these fixtures pin the C++ oracle and the HIP kernels to
a second, structurally different statement of the same formulas on the reference's own test elements
(tests/LocalOperatorCommon.hpp:17-59) with fixed seeds.  Inputs and expected outputs only -- no reference text.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_np as P  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
# tests/LocalOperatorCommon.hpp:28-29, 47-56
QUAD = np.array([[1, 1, 0], [2, 1, 0], [1, 3, 0], [3, 4, 0]], float)
HEX = np.array([[1, 1, 0], [2, 1, 0], [1, 3, 0], [3, 4, 0], [1, 1, 1], [2, 1, 1.5], [1, 3, 2], [3, 4, 3.5]], float)


def tables():
    d = {}
    for p in range(1, 9):
        for nq in sorted({p + 1, 2 * p + 1}):
            gll, qx, qw, I, D = P.tables(p, nq)
            d[f"gll_{p}"] = gll
            d[f"qx_{nq}"], d[f"qw_{nq}"] = qx, qw
            d[f"I_{p}_{nq}"], d[f"D_{p}_{nq}"] = I, D
    np.savez_compressed(os.path.join(OUT, "tables.npz"), **d)


def element_case(name, kid, p, nq, R, verts, seed, kparams=None, store_K=True, dirichlet=True, time=0.0):
    dim, E, U, F = P.kernel_params(kid)
    rng = np.random.default_rng(seed)
    N = (p + 1) ** dim
    nf = rng.uniform(-1, 1, (N, F)) if F else None
    K, Fe = P.assemble(kid, p, nq, R, verts, nf, kparams, time)
    x = rng.uniform(-1, 1, (N * U, R))
    d = dict(kid=kid, p=p, nq=nq, R=R, verts=verts, x=x, y=K @ x, F=Fe, diag=np.diag(K).copy(), time=time)
    if kparams is not None:
        d["kparams"] = np.asarray(kparams, float)
    if nf is not None:
        d["node_fields"] = nf
    if store_K:
        d["K"] = K
    if dirichlet:  # Dirichlet on unknown 0 at the boundary nodes, random values: lifted rhs = F - K[:,D] g
        bn = P.boundary_nodes(dim, p)
        dinds = bn * U
        g = rng.uniform(-1, 1, (len(dinds), R))
        d["dir_inds"], d["dir_vals"] = dinds.astype(np.int32), g
        d["rhs_lifted"] = Fe - K[:, dinds] @ g
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "Nd", N * U, "|K|max", np.abs(K).max())


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--only-new" in sys.argv:  # the fixtures of earlier rounds stay byte-identical
        _existing = set(os.listdir(OUT))
        _all_case = element_case
        element_case = lambda name, *a, **k: None if name + ".npz" in _existing else _all_case(name, *a, **k)  # noqa: E731
    else:
        tables()
    element_case("hex_p3_diff", 0, 3, 7, 3, HEX, 1, kparams=[1.0, 0.0])       # K1/K2 element (value_order=2)
    element_case("hex_p3_var", 1, 3, 7, 2, HEX, 2)                              # K3 element
    element_case("quad_p4_diff", 2, 4, 9, 2, QUAD, 3)
    element_case("quad_p4_var", 3, 4, 9, 2, QUAD, 4)
    element_case("hex_p4_diff", 0, 4, 5, 1, HEX, 5, kparams=[1.0, 1.0], store_K=False)  # config 2 shape
    element_case("hex_p6_diff", 0, 6, 7, 1, HEX, 6, kparams=[1.0, 1.0], store_K=False)  # north-star shape
    element_case("hex_p4_advdiff", 4, 4, 5, 1, HEX, 7, kparams=[0.7, 1.3, 0.5], store_K=False)  # config 5 shape
    element_case("hex_p2_advdiff", 4, 2, 3, 2, HEX, 8, kparams=[0.7, 1.3, 0.5])
    # round 4: kernels that read the space-time point (true z), odd numbers of unknowns, the reference's NS3D benchmark kernel
    element_case("hex_p2_point", 10, 2, 5, 1, HEX, 9, kparams=[0.8, 1.2], time=0.7)     # value_order = 2
    element_case("hex_p4_point", 10, 4, 5, 1, HEX, 10, kparams=[0.8, 1.2], time=-0.4, store_K=False)
    element_case("hex_p4_advection", 11, 4, 5, 1, HEX, 11, kparams=[0.05])
    element_case("hex_p2_divcurl", 12, 2, 3, 1, HEX, 12, kparams=[0.6])
    element_case("hex_p2_ns3d", 13, 2, 4, 1, HEX, 13)                                    # QO = 4p - 1 -> nq = 2p
