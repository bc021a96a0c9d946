"""The reference's own heavy kernel through the kept kernel-definition API: NS3D, the linearised Navier-Stokes kernel of its
micro-benchmarks (benchmarks/Kernels.hpp:3-65: U = 7 unknowns, E = 8 equations, F = 7 fields read by value AND derivative;
benchmarks/LocalAssemblyBenchmarks.cpp:42-87, LocalOperatorEvaluationBenchmarks.cpp:3-47: QO = 4p - 1, i.e. nq = 2p), as a run-time
PLUGIN: tests/kernels/ns3d.hpp is the body of the reference's lambda with L3K_HD added -- structured bindings over the seven field
values and their derivative arrays included.  Checked against the oracle's restatement of the same lambda (kernel 13) and the numpy
restatement's fixture: apply on both routes, diag / lifted rhs, K_e / F_e entry by entry, at p = 2 (nq = 4) and p = 4 (nq = 8).
At p = 4 the element's buffers exceed the LDS (14 fields x 8^3 points x 5 buffers = 286 KB): the generic kernels then run on a
global-memory working set (sumfact_apply.hpp GS) -- the route line says so."""
import os

import numpy as np
import pytest

import oracle_lib as O
from helpers import SingleElementMesh, oracle_mesh, rel_err

KID = 1013
# the benchmarks fix QO = 4p - 1, i.e. nq = 2p points per direction; through the options: value_order 1, derivative_order 1
# (nq = value_order * p + derivative_order * (p - 1) + 1, algsys/AssembleLocalSystem.hpp:32-35)
OPTS = (1, 1, 0)
SOURCE = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernels", "ns3d.hpp")).read()


@pytest.fixture(scope="module")
def ns3d():
    from l3ster_amd import plugin
    return plugin.compile_kernel("NS3D", SOURCE, KID, shapes=[(2, 4, 1), (4, 8, 1), (6, 12, 1)])


def test_plugin_builds_and_registers(ns3d):
    from l3ster_amd import system
    info = system.kernel_info(ns3d)
    assert (info["n_equations"], info["n_unknowns"], info["n_fields"], info["param_bytes"]) == (8, 7, 7, 0)
    assert all((KID, p, 2 * p, 1) in system.instances() for p in (2, 4, 6))
    assert O.kernel_params(O.KERNEL_NS3D) == dict(dim=3, E=8, U=7, F=7)


def _ctx():
    import torch
    from l3ster_amd import system
    torch.cuda.set_device(0)
    return system.Context(0, torch.cuda.current_stream().cuda_stream)


def _dev(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")


@pytest.mark.gpu
def test_single_element_vs_golden(ns3d, golden):
    import torch
    from l3ster_amd import system
    g = golden("hex_p2_ns3d")
    p, nq, U = int(g["p"]), int(g["nq"]), 7
    assert nq == 2 * p and system.n_qps1d(p, *OPTS[:2]) == nq
    ctx = _ctx()
    Nd = (p + 1) ** 3 * U
    mask = np.zeros(Nd, np.uint8)
    mask[g["dir_inds"]] = 1
    for route, below in (("fast", 0), ("generic", 10 ** 9)):
        with ctx.tuning(generic_below=below):
            mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, SingleElementMesh(p, g["verts"]), U), ns3d, asm_opts=OPTS)
            mf.set_fields(_dev(g["node_fields"].T))
            assert mf.route().startswith("sumfactFastKernel<p=2,nq=4,U=7,F=7>" if route == "fast" else "sumfactApplyKernel<p=2,nq=4,U=7,F=7"), mf.route()
            X, Y = _dev(g["x"].T), torch.zeros((1, Nd), dtype=torch.float64, device="cuda")
            mf.apply(X, Y)
            assert rel_err(Y.cpu().numpy().T, g["y"]) < 1e-12, route
    K, Fe, _ = mf.local_assemble()
    assert np.abs(K.cpu().numpy()[0] - g["K"]).max() < 1e-12 * np.abs(g["K"]).max()
    assert np.abs(Fe.cpu().numpy()[0].T - g["F"]).max() < 1e-12 * max(1.0, np.abs(g["F"]).max())
    mfd = system.MatrixFreeSystem(system.DeviceMesh(ctx, SingleElementMesh(p, g["verts"]), U, mask), ns3d, asm_opts=OPTS)
    mfd.set_fields(_dev(g["node_fields"].T))
    gd = np.zeros((1, Nd))
    gd[:, g["dir_inds"]] = g["dir_vals"].T
    diag, rhs = mfd.diag_rhs(_dev(gd), finalize=False)
    np.testing.assert_allclose(diag.cpu().numpy(), g["diag"], rtol=1e-12, atol=1e-13)
    free = mask == 0
    assert rel_err(rhs.cpu().numpy().T[free], g["rhs_lifted"][free]) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("p", [2, 4, 6])
def test_mesh_vs_oracle(ns3d, p):
    """apply (alpha, beta; Dirichlet on the three velocity components of all sides), diag and lifted rhs on a perturbed 2^3 mesh
    (2 x 1 x 1 at p = 6), and K_e / F_e of two of its elements entry by entry, against the oracle (p = 6: F_e and the checksum
    only -- the oracle's dense 2401 x 2401 K_e costs 80 GFLOP per element on the CPU)."""
    import torch
    from l3ster_amd import system
    ctx = _ctx()
    U, F, nq = 7, 7, 2 * p
    part = system.CubePartition(2 if p < 6 else (2, 1, 1), p, perturb=0.15)
    mask = part.dirichlet_mask(U, unknowns=(0, 1, 2))
    fields = np.random.default_rng(4).uniform(-1, 1, (F, part.n_local_nodes))
    assert system.n_qps1d(p, *OPTS[:2]) == nq
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U, mask), ns3d, asm_opts=OPTS)
    mf.set_fields(_dev(fields))
    om = oracle_mesh(part, nq, U, np.arange(U), mask, fields)
    x = part.synthetic_vector(U)
    y0 = np.random.default_rng(1).uniform(-1, 1, x.shape)
    gd = np.random.default_rng(6).uniform(-1, 1, (1, part.n_local_nodes * U)) * mask[None, :]
    y_ref = O.mf_apply(om, O.KERNEL_NS3D, x.T, np.asfortranarray(y0.T.copy()), alpha=1.5, beta=-0.25)
    d_ref, r_ref = O.mf_diag_rhs(om, O.KERNEL_NS3D, 1, np.asfortranarray(gd.T))
    routes = [("generic", 10 ** 9)] + ([("fast", 0)] if p == 2 else [])
    for route, below in routes:
        with ctx.tuning(generic_below=below):
            line = mf.route()
            if p >= 4:  # 286 KB (p = 4) / 968 KB (p = 6) of buffers per element: the global-scratch variant of the generic kernel
                assert "GLOBAL scratch" in line and f"sumfactApplyKernel<p={p},nq={2 * p},U=7,F=7" in line, line
            X, Y = _dev(x), _dev(y0)
            mf.apply(X, Y, 1.5, -0.25)
            assert rel_err(Y.cpu().numpy().T, y_ref) < 1e-11, line
            diag, rhs = mf.diag_rhs(_dev(gd))
            assert rel_err(diag.cpu().numpy(), d_ref) < 1e-11 and rel_err(rhs.cpu().numpy().T, r_ref) < 1e-11, line
    K, Fe, cs = mf.local_assemble(0, 2, want_checksum=True)
    _, _, cs_stream = mf.local_assemble(0, 2, want_K=False, want_F=False, want_checksum=True)
    K, Fe = K.cpu().numpy(), Fe.cpu().numpy()
    np.testing.assert_allclose(cs_stream.cpu().numpy(), cs.cpu().numpy(), rtol=1e-11)
    if p == 6:  # F_e through the oracle's lifted-rhs function without Dirichlet dofs; K_e through its action on a vector
        for e in range(2):
            nf = fields[:, part.elem_nodes[e]].T
            _, F_ref = O.diag_rhs_local(O.KERNEL_NS3D, p, nq, 1, part.elem_verts[e], None, None, nf)
            assert np.abs(Fe[e].T - F_ref).max() < 1e-12 * max(1.0, np.abs(F_ref).max()), e
            v = np.random.default_rng(e).uniform(-1, 1, (K.shape[1], 1))
            assert rel_err(K[e] @ v, O.apply_local(O.KERNEL_NS3D, p, nq, part.elem_verts[e], v, nf)) < 1e-12, e
        return
    for e in range(2):
        K_ref, F_ref = O.assemble_local(O.KERNEL_NS3D, p, nq, 1, part.elem_verts[e], fields[:, part.elem_nodes[e]].T)
        assert np.abs(K[e] - K_ref).max() < 1e-12 * np.abs(K_ref).max(), e
        assert np.abs(Fe[e].T - F_ref).max() < 1e-12 * max(1.0, np.abs(F_ref).max()), e
