"""GPU parity tests of LocalAssembly (K_e, F_e: the sum-factorised assembly kernel on the FP64 vector pipe by default, the
dense (W Z)^T Z product on the FP64 matrix cores behind l3k_tuning::assemble_dense): HIP path through the C ABI vs the oracle's
assembleLocalSystem restatement and the golden fixtures.  Tolerance: max-norm 1e-12 relative to |K_e|_max (stated fp64
tolerance)."""
import numpy as np
import pytest

import oracle_lib as O
from helpers import HEX, SingleElementMesh
from l3ster_amd import system

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    return system.Context(0, torch.cuda.current_stream().cuda_stream)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")


def checksum_of(K):
    n = K.shape[0]
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    return float((K * (1 + (i * 31 + j * 17) % 7)).sum())


CASES = [
    # kid, ne, p, value_order, R, kparams
    (system.KERNEL_DIFFUSION3D, 2, 1, 1, 1, [0.7, 1.3]),
    (system.KERNEL_DIFFUSION3D, 2, 2, 1, 2, [0.7, 1.3]),
    (system.KERNEL_DIFFUSION3D, (2, 1, 1), 3, 2, 3, [1.0, 0.5]),
    (system.KERNEL_DIFFUSION3D_VAR, (2, 1, 1), 3, 2, 2, None),
    (system.KERNEL_ADVDIFF3D, 2, 2, 1, 2, [0.7, 1.3, 0.5]),
    (system.KERNEL_DIFFUSION3D, (2, 1, 1), 4, 1, 1, [1.0, 1.0]),
]


@pytest.mark.parametrize("kid,ne,p,vo,R,kpar", CASES)
def test_local_assembly_vs_oracle(ctx, kid, ne, p, vo, R, kpar):
    info = system.kernel_info(kid)
    U, F = info["n_unknowns"], info["n_fields"]
    part = system.CubePartition(ne, p, perturb=0.15)
    nq = system.n_qps1d(p, vo)
    mesh = system.DeviceMesh(ctx, part, U)
    mf = system.MatrixFreeSystem(mesh, kid, kpar, asm_opts=(vo, 0, 0), n_rhs=R)
    fields = np.random.default_rng(2).uniform(-1, 1, (F, part.n_local_nodes)) if F else None
    if F:
        mf.set_fields(dev(fields))
    K, Fe, cs = mf.local_assemble(want_checksum=True)
    torch.cuda.synchronize()
    K, Fe, cs = K.cpu().numpy(), Fe.cpu().numpy(), cs.cpu().numpy()
    for e in range(part.n_elems):
        nf = fields[:, part.elem_nodes[e]].T if F else None
        K_ref, F_ref = O.assemble_local(kid, p, nq, R, part.elem_verts[e], nf, kpar)
        scale = np.abs(K_ref).max()
        assert np.abs(K[e] - K_ref).max() < 1e-12 * scale
        assert np.abs(Fe[e].T - F_ref).max() < 1e-12 * max(1.0, np.abs(F_ref).max())
        assert np.array_equal(K[e], K[e].T)  # mirrored from the lower triangle (AssembleLocalSystem.hpp:176-182)
        assert abs(cs[e] - checksum_of(K[e])) < 1e-9 * abs(checksum_of(np.abs(K[e])))


@pytest.mark.parametrize("kid,ne,p,vo,R,kpar", CASES + [
    (system.KERNEL_DIFFUSION3D, 1, 5, 1, 1, [0.7, 1.3]),   # n = 6: the middle b_x' of the diagonal blocks stands alone
    (system.KERNEL_DIFFUSION3D, 1, 6, 1, 1, [0.7, 1.3]),
    (system.KERNEL_DIFFUSION3D, 1, 7, 1, 1, [1.0, 0.5]),   # n = 8; the dense kernel unless the sum-factorised one fits
    (system.KERNEL_ADVDIFF3D, (2, 1, 1), 4, 1, 1, [0.7, 1.3, 0.5]),  # external fields + the DPP kernels (orders >= 4)
])
def test_diagonal_block_kernel_vs_oracle(ctx, kid, ne, p, vo, R, kpar):
    """The streaming mode forms the diagonal blocks (u' == u) by halves in merged iterations, in a kernel of its own
    (device/assemble.hpp, BLOCKS == 1 / 2).  l3k_tuning::assemble_two_launches makes the stored mode take the same two kernels: K_e entry by
    entry against the oracle, bitwise symmetric; and the streaming checksum equals the stored matrix's."""
    info = system.kernel_info(kid)
    U, F = info["n_unknowns"], info["n_fields"]
    part = system.CubePartition(ne, p, perturb=0.15)
    nq = system.n_qps1d(p, vo)
    mesh = system.DeviceMesh(ctx, part, U)
    mf = system.MatrixFreeSystem(mesh, kid, kpar, asm_opts=(vo, 0, 0), n_rhs=R)
    fields = np.random.default_rng(2).uniform(-1, 1, (F, part.n_local_nodes)) if F else None
    if F:
        mf.set_fields(dev(fields))
    with ctx.tuning(assemble_two_launches=1):
        K, _, cs = mf.local_assemble(want_checksum=True)
    _, _, cs_stream = mf.local_assemble(want_K=False, want_F=False, want_checksum=True)
    K1, _, _ = mf.local_assemble()
    torch.cuda.synchronize()
    K, K1, cs, cs_stream = K.cpu().numpy(), K1.cpu().numpy(), cs.cpu().numpy(), cs_stream.cpu().numpy()
    for e in range(part.n_elems):
        nf = fields[:, part.elem_nodes[e]].T if F else None
        K_ref, _ = O.assemble_local(kid, p, nq, R, part.elem_verts[e], nf, kpar)
        scale = np.abs(K_ref).max()
        assert np.abs(K[e] - K_ref).max() < 1e-12 * scale
        assert np.abs(K1[e] - K_ref).max() < 1e-12 * scale
        assert np.array_equal(K[e], K[e].T)
        ref_cs = checksum_of(np.abs(K[e]))
        assert abs(cs[e] - checksum_of(K[e])) < 1e-9 * ref_cs
        assert abs(cs_stream[e] - checksum_of(K[e])) < 1e-9 * ref_cs


def test_local_assembly_order6_vs_golden(ctx, golden):
    """The north-star shape (order 6, 1372 x 1372) on the reference's distorted test hex: K x, diag(K), F vs the
    independent numpy/mpmath restatement; checksum-only (streaming) mode gives the same checksum."""
    g = golden("hex_p6_diff")
    mesh = system.DeviceMesh(ctx, SingleElementMesh(6, g["verts"]), 4)
    mf = system.MatrixFreeSystem(mesh, system.KERNEL_DIFFUSION3D, g["kparams"])
    K, Fe, cs = mf.local_assemble(want_checksum=True)
    _, _, cs2 = mf.local_assemble(want_K=False, want_F=False, want_checksum=True)
    torch.cuda.synchronize()
    K = K.cpu().numpy()[0]
    assert np.linalg.norm(K @ g["x"] - g["y"]) < 1e-12 * np.linalg.norm(g["y"])
    np.testing.assert_allclose(np.diag(K), g["diag"], rtol=1e-12)
    np.testing.assert_allclose(Fe.cpu().numpy()[0].T, g["F"], atol=1e-13)
    assert np.array_equal(K, K.T) and np.linalg.eigvalsh(K).min() > -1e-10
    assert abs(cs.item() - cs2.item()) <= 1e-12 * abs(cs.item())
    assert abs(cs.item() - checksum_of(K)) < 1e-9 * checksum_of(np.abs(K))


@pytest.mark.parametrize("dense", [False, True])
def test_local_assembly_order6_vs_oracle_entrywise(ctx, dense):
    """The north-star assembly shape entry by entry: the full 1372 x 1372 K_e and F_e of the default sum-factorised kernel
    and of the dense MFMA product (l3k_tuning::assemble_dense) against the oracle's assembleLocalSystem
    (algsys/AssembleLocalSystem.hpp:77-216,234-256) on the reference's distorted test hex and on elements of a perturbed
    mesh; max-norm 1e-12 relative to |K_e|_max."""
    p, U, kpar, kid = 6, 4, [0.7, 1.3], system.KERNEL_DIFFUSION3D
    cases = [(SingleElementMesh(p, HEX), [0]), (system.CubePartition((2, 2, 1), p, perturb=0.2), [0, 3])]
    for part, elems in cases:
        mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U), kid, kpar)
        with ctx.tuning(assemble_dense=int(dense)):
            K, Fe, _ = mf.local_assemble()
        torch.cuda.synchronize()
        K, Fe = K.cpu().numpy(), Fe.cpu().numpy()
        for e in elems:
            K_ref, F_ref = O.assemble_local(kid, p, p + 1, 1, part.elem_verts[e], None, kpar)
            scale = np.abs(K_ref).max()
            assert K[e].shape == (1372, 1372)
            assert np.abs(K[e] - K_ref).max() < 1e-12 * scale
            assert np.abs(Fe[e].T - F_ref).max() < 1e-12 * max(1.0, np.abs(F_ref).max())
            assert np.array_equal(K[e], K[e].T)


@pytest.mark.parametrize("kid,ne,p,vo,kpar", [(system.KERNEL_DIFFUSION3D, 2, 2, 1, [0.7, 1.3]), (system.KERNEL_DIFFUSION3D, (2, 1, 1), 3, 2, [1.0, 0.5]),
                                              (system.KERNEL_MASS3D, 2, 3, 2, None), (system.KERNEL_DIFFUSION3D, (2, 1, 1), 4, 1, [1.0, 1.0]),
                                              (system.KERNEL_DIFFUSION3D, 1, 6, 1, [0.7, 1.3])])
def test_tiled_layout_is_the_row_major_matrix(ctx, kid, ne, p, vo, kpar):
    """l3k_local_assemble_tiled (every U x U block, stored [u][u'][bx'][bz][bx][by][by'][bz']: the layout l3k_assemble_global keeps
    between its kernels) holds the entries of the row-major K_e of l3k_local_assemble -- which the tests above pin against the
    oracle -- including the mirrored halves, which the tiled kernel computes instead of copying."""
    U = system.kernel_info(kid)["n_unknowns"]
    part = system.CubePartition(ne, p, perturb=0.15)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U), kid, kpar, asm_opts=(vo, 0, 0))
    K, _, _ = mf.local_assemble(want_F=False)
    Kt = mf.local_assemble_tiled()
    torch.cuda.synchronize()
    n = p + 1
    # [e, u, u', bx', bz, bx, by, by', bz'] -> [e, (bz, by, bx, u), (bz', by', bx', u')]
    K2 = Kt.permute(0, 4, 6, 5, 1, 8, 7, 3, 2).reshape(part.n_elems, n ** 3 * U, n ** 3 * U)
    scale = K.abs().amax()
    assert float((K2 - K).abs().amax()) < 1e-13 * float(scale)


def test_degenerate_element_is_an_error(ctx):
    bad = HEX.copy()
    bad[[0, 1]] = bad[[1, 0]]
    mesh = system.DeviceMesh(ctx, SingleElementMesh(2, bad), 4)
    mf = system.MatrixFreeSystem(mesh, system.KERNEL_DIFFUSION3D)
    with pytest.raises(system.L3KError, match="degenerate"):
        mf.local_assemble()


@pytest.mark.parametrize("p,vo", [(3, 2), (2, 1)])
def test_mass_kernel_pins_weight_times_jacobian(ctx, p, vo):
    """Device twin of tests/test_reference_kats_tables.py::test_mass_kernel_pins_weight_times_jacobian: A0 = I on the
    reference's distorted hex (tests/LocalOperatorCommon.hpp:36-59): sum_ij K_e[(i,u),(j,u)] = volume = 22/3, off-diagonal
    unknown blocks vanish, sum_i F_e[(i,u)] = rhs_u * volume -- through l3k_local_assemble (the sum-factorised kernels) and through the
    matrix-free apply y = A * 1.  An error in the per-point weight w * detJ shows in every one of these numbers."""
    vol = 22.0 / 3.0
    mesh = system.DeviceMesh(ctx, SingleElementMesh(p, HEX), 2)
    mf = system.MatrixFreeSystem(mesh, system.KERNEL_MASS3D, asm_opts=(vo, 0, 0))
    K, Fe, _ = mf.local_assemble()
    torch.cuda.synchronize()
    K, Fe = K.cpu().numpy()[0], Fe.cpu().numpy()[0]
    assert abs(K[0::2, 0::2].sum() - vol) < 1e-11 and abs(K[1::2, 1::2].sum() - vol) < 1e-11
    assert np.abs(K[0::2, 1::2]).max() == 0.0
    np.testing.assert_allclose([Fe[0, 0::2].sum(), Fe[0, 1::2].sum()], [vol, 2 * vol], atol=1e-11)
    K_ref, F_ref = O.assemble_local(system.KERNEL_MASS3D, p, system.n_qps1d(p, vo), 1, HEX)
    assert np.abs(K - K_ref).max() < 1e-12 * np.abs(K_ref).max()
    ones = np.zeros(K.shape[0])
    ones[0::2] = 1.0
    for generic_below in (0, 1000000):  # the one-wave kernel and the generic LDS kernel
        with ctx.tuning(generic_below=generic_below):
            assert ("sumfactApplyKernel" if generic_below else "sumfactFastKernel") in mf.route()
            X, Y = dev(ones[None, :]), dev(np.zeros((1, K.shape[0])))
            mf.apply(X, Y, 1.0, 0.0)
            torch.cuda.synchronize()
        y = Y.cpu().numpy()[0]
        np.testing.assert_allclose(y, K_ref @ ones, atol=1e-12)
        assert abs(y[0::2].sum() - vol) < 1e-11 and np.abs(y[1::2]).max() < 1e-13


def test_config3_full_streaming_sweep_64cubed_order6(ctx):
    """BASELINE.json configs[2] as stated: Diffusion3D, hex mesh 64^3, order 6, LocalAssembly (K_e = sum_q w detJ B^T B; the
    default sum-factorised kernel) over the WHOLE mesh in streaming mode (the 262 144 matrices would be 3.9 TB: each K_e is reduced to its weighted
    checksum sum_ij K_ij (1 + (31 i + 17 j) mod 7) on the device).  32 elements drawn at random over the mesh are checked
    against the oracle: with c_ij = 1 + 3 (i + j) mod 7 the checksum is sum_b sum_i (K E)_ib (1 + 3 (i + b) mod 7) for the
    seven indicator columns E_b = [j = b mod 7], i.e. seven applications of the oracle's element operator."""
    import time
    p, U, kpar = 6, 4, [1.0, 1.0]
    part = system.CubePartition(64, p, perturb=0.1)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U), system.KERNEL_DIFFUSION3D, kpar)
    batch = 512
    cs = torch.empty(part.n_elems, dtype=torch.float64, device="cuda")
    t0 = time.perf_counter()
    for first in range(0, part.n_elems, batch):
        _, _, c = mf.local_assemble(first, batch, want_K=False, want_F=False, want_checksum=True)
        cs[first:first + batch] = c
    torch.cuda.synchronize()
    rate = part.n_elems / (time.perf_counter() - t0)
    cs = cs.cpu().numpy()
    # (~520 k element matrices/s measured for the sum-factorised kernels in this loop with its per-batch host synchronisation, 9.8 k
    # for the dense MFMA product: a regression to the dense route or a 2x slowdown of the default kernels fails)
    assert np.all(np.isfinite(cs)) and rate > 250_000, rate
    Nd = (p + 1) ** 3 * U
    E = np.zeros((Nd, 7))
    E[np.arange(Nd), np.arange(Nd) % 7] = 1.0
    wgt = 1.0 + (3 * (np.arange(Nd)[:, None] + np.arange(7)[None, :])) % 7
    sample = np.random.default_rng(11).choice(part.n_elems, 32, replace=False)
    for e in sample:
        Y = O.apply_sumfact(system.KERNEL_DIFFUSION3D, p, p + 1, part.elem_verts[e], E, kparams=kpar)
        ref = float((Y * wgt).sum())
        assert abs(cs[e] - ref) < 1e-10 * float((np.abs(Y) * wgt).sum()), (e, cs[e], ref)


def test_config3_stored_row_major_order6_vs_oracle_and_rate(ctx):
    """Config 3's matrices as the reference returns them -- row-major K_e stored in HBM, symmetric bit for bit -- for 128 elements of
    the 64^3 order-6 mesh in one call (x-major tiled assembly of the lower triangle + mirroring transposition, two sub-batches): two
    matrices entrywise against the oracle's local-element assembly, all of them bitwise symmetric, and a floor on the rate (98 k
    matrices/s measured; the round's first route gave 66 k, the assembly kernel's direct store 42 k)."""
    p, U, kpar, n = 6, 4, [1.0, 1.0], 128
    part = system.CubePartition(64, p, perturb=0.1)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U), system.KERNEL_DIFFUSION3D, kpar)
    Nd = (p + 1) ** 3 * U
    first = 70_000
    K = torch.empty((n, Nd, Nd), dtype=torch.float64, device="cuda")
    mf.local_assemble_into(K, first, n)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(3):
        mf.local_assemble_into(K, first, n)
    t1.record()
    torch.cuda.synchronize()
    rate = 3 * n / (t0.elapsed_time(t1) * 1e-3)
    assert torch.equal(K, K.transpose(1, 2))
    for i in (0, 101):  # (one element of each sub-batch)
        Kref, _ = O.assemble_local(system.KERNEL_DIFFUSION3D, p, p + 1, 1, part.elem_verts[first + i], kparams=kpar)
        got = K[i].cpu().numpy()
        assert np.abs(got - Kref).max() < 1e-12 * np.abs(Kref).max()
    assert rate > 60_000, rate


def _csr_graph(part, dpn, field_inds):
    """CSR graph of the rank-local matrix: row / column dofs of every element coupled (what the reference's sparsity graph
    holds for one domain kernel), columns ascending -- built on the host like the caller's Tpetra graph would be."""
    import scipy.sparse as sp
    dofs = (part.elem_nodes.astype(np.int64)[:, :, None] * dpn + np.asarray(field_inds)[None, None, :]).reshape(part.n_elems, -1)
    nd = dofs.shape[1]
    rows = np.repeat(dofs, nd, axis=1).ravel()
    cols = np.tile(dofs, (1, nd)).ravel()
    n = part.n_local_nodes * dpn
    G = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n)).tocsr()
    G.sort_indices()
    return G.indptr.astype(np.int64), G.indices.astype(np.int32), n


@pytest.mark.parametrize("mode", ["node_rows", "per_entry", "global"])
@pytest.mark.parametrize("kid,p,vo,R,kpar", [(system.KERNEL_DIFFUSION3D, 2, 1, 2, [0.7, 1.3]), (system.KERNEL_MASS3D, 3, 2, 1, None),
                                             (system.KERNEL_DIFFUSION3D, 4, 1, 1, [1.0, 1.0])])
def test_assembled_scatter_vs_oracle_dense(ctx, kid, p, vo, R, kpar, mode, request):
    """a20, scatterLocalSystem + assembleGlobalSystem (algsys/ScatterLocalSystem.hpp:24-54, AssembleGlobalSystem.hpp:20-53)
    on the device: local systems of a 3^3 (2^3 at order 4) distorted mesh from l3k_local_assemble summed into the caller's
    CSR values and the global right-hand sides, in two batches; against the oracle's element systems added into a dense
    global matrix on the host.  Then with skip_dirichlet: the assembled operator equals the matrix-free apply."""
    import scipy.sparse as sp
    # node_rows (default): one wave per (element, row node), one search per entry shared by the node's U rows; per_entry
    # (l3k_tuning::scatter_per_entry): the round-2 kernel, a search per entry; global: l3k_assemble_global, the whole pipeline inside
    # the library (sub-batches of element systems formed on one stream and scattered on another)
    ctx.set_tuning(scatter_per_entry=int(mode == "per_entry"))
    request.addfinalizer(lambda: ctx.set_tuning(scatter_per_entry=0))
    info = system.kernel_info(kid)
    U = info["n_unknowns"]
    ne = 2 if p == 4 else 3
    part = system.CubePartition(ne, p, perturb=0.15)
    nq = system.n_qps1d(p, vo)
    mask = part.dirichlet_mask(U) if U == 4 else None
    mesh = system.DeviceMesh(ctx, part, U, mask)
    mf = system.MatrixFreeSystem(mesh, kid, kpar, asm_opts=(vo, 0, 0), n_rhs=R)
    row_ptr, col_ind, n = _csr_graph(part, U, np.arange(U))
    RP, CI = torch.as_tensor(row_ptr, device="cuda"), torch.as_tensor(col_ind, device="cuda")

    def assemble(skip):
        vals = torch.zeros(len(col_ind), dtype=torch.float64, device="cuda")
        rhs = torch.zeros((R, n), dtype=torch.float64, device="cuda")
        half = part.n_elems // 2
        if mode == "global":
            Nd = (p + 1) ** 3 * U
            ws = 2 * 3 * 8 * (Nd * Nd + Nd * R + 8 * (p + 2) ** 3 * 128)  # room for about three elements per half: many sub-batches
            assert mf.assemble_global(RP, CI, vals, rhs, first=0, count=half, skip_dirichlet=skip, workspace_bytes=ws) == 0
            assert mf.assemble_global(RP, CI, vals, rhs, first=half, skip_dirichlet=skip) == 0
        else:
            for first, count in ((0, half), (half, part.n_elems - half)):
                K, Fe, _ = mf.local_assemble(first, count)
                assert mf.assembled_scatter(K, Fe, RP, CI, vals, rhs, first=first, skip_dirichlet=skip) == 0
        torch.cuda.synchronize()
        return sp.csr_matrix((vals.cpu().numpy(), col_ind, row_ptr), shape=(n, n)), rhs.cpu().numpy()

    A, rhs = assemble(False)
    A_ref, rhs_ref = np.zeros((n, n)), np.zeros((R, n))
    for e in range(part.n_elems):
        K_ref, F_ref = O.assemble_local(kid, p, nq, R, part.elem_verts[e], None, kpar)
        dofs = (part.elem_nodes[e].astype(np.int64)[:, None] * U + np.arange(U)[None, :]).ravel()
        A_ref[np.ix_(dofs, dofs)] += K_ref
        rhs_ref[:, dofs] += F_ref.T
    assert np.abs(A.toarray() - A_ref).max() < 1e-12 * np.abs(A_ref).max()
    assert np.abs(rhs - rhs_ref).max() < 1e-12 * max(1.0, np.abs(rhs_ref).max())
    if kid == system.KERNEL_MASS3D:  # the global mass matrix sums to the volume of the (perturbed) unit cube per unknown
        assert abs(A.toarray()[0::2, 0::2].sum() - 1.0) < 1e-12
    # a graph that lacks entries: they are skipped and counted (sumIntoLocalValues semantics)
    keep = np.ones(len(col_ind), bool)
    keep[row_ptr[5]:row_ptr[6]][::2] = False
    rp2 = np.concatenate([[0], np.cumsum(np.add.reduceat(keep.astype(np.int64), row_ptr[:-1]))])
    vals2 = torch.zeros(int(keep.sum()), dtype=torch.float64, device="cuda")
    K, _, _ = mf.local_assemble(0, part.n_elems)
    missing = mf.assembled_scatter(K, None, torch.as_tensor(rp2, device="cuda"), torch.as_tensor(col_ind[keep], device="cuda"), vals2, None)
    assert missing > 0
    A2 = sp.csr_matrix((vals2.cpu().numpy(), col_ind[keep], rp2), shape=(n, n)).toarray()
    dropped = np.zeros((n, n), bool)
    dropped[5, col_ind[row_ptr[5]:row_ptr[6]][::2]] = True
    assert np.abs(np.where(dropped, 0.0, A_ref) - A2).max() < 1e-12 * np.abs(A_ref).max()
    if mask is not None:
        # assembled == matrix-free on the free dofs (the reference's cross-path property, tests/LocalOperatorTests.cpp:3-95,
        # at mesh level): Dirichlet rows / columns left out of the sum, y[D] = x[D] added by hand
        Af, _ = assemble(True)
        x = part.synthetic_vector(U)
        X, Y = dev(x), dev(np.zeros_like(x))
        mf.apply(X, Y, 1.0, 0.0)
        torch.cuda.synchronize()
        want = Af @ x[0] + np.where(mask, x[0], 0.0)
        assert np.linalg.norm(Y.cpu().numpy()[0] - want) < 1e-11 * np.linalg.norm(want)


@pytest.mark.parametrize("mode", ["node_rows", "per_entry"])
def test_rhs_scatter_with_more_than_64_rhs_entries_per_node(ctx, mode):
    """the per-row-node scatter kernels added F_e with `lane < n_rhs * U` in one 64-lane wave, so with n_rhs * U > 64
    (here 17 right-hand sides of 4 unknowns) the remaining entries were silently dropped.  F_e only (scatterLocalSystem's
    atomic rhs adds, algsys/ScatterLocalSystem.hpp:45-52), against a host scatter-add; with skipped Dirichlet rows too."""
    U, p, R = 4, 2, 17
    part = system.CubePartition(3, p, perturb=0.1)
    mask = part.dirichlet_mask(U)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U, mask), system.KERNEL_DIFFUSION3D, n_rhs=R)
    Nd, n = (p + 1) ** 3 * U, part.n_local_nodes * U
    F = np.random.default_rng(5).uniform(-1, 1, (part.n_elems, R, Nd))
    dofs = (part.elem_nodes.astype(np.int64)[:, :, None] * U + np.arange(U)[None, None, :]).reshape(part.n_elems, Nd)
    with ctx.tuning(scatter_per_entry=int(mode == "per_entry")):
        for skip in (False, True):
            want = np.zeros((R, n))
            for r in range(R):
                np.add.at(want[r], dofs.reshape(-1), F[:, r, :].reshape(-1))
            if skip:
                want[:, mask.astype(bool)] = 0.0
            rhs = torch.zeros((R, n), dtype=torch.float64, device="cuda")
            mf.assembled_scatter(None, dev(F), None, None, None, rhs, skip_dirichlet=skip)
            np.testing.assert_allclose(rhs.cpu().numpy(), want, rtol=0, atol=1e-13)


@pytest.mark.parametrize("kid,ne,p,vo,kpar", [(system.KERNEL_DIFFUSION3D, (3, 2, 2), 2, 1, [0.7, 1.3]), (system.KERNEL_DIFFUSION3D, (2, 2, 1), 4, 1, [1.0, 1.0]),
                                              (system.KERNEL_DIFFUSION3D, (2, 1, 1), 6, 1, [0.7, 1.3]), (system.KERNEL_MASS3D, 2, 3, 2, None),
                                              (system.KERNEL_DIVCURL3D, (2, 1, 1), 4, 1, [0.6]), (system.KERNEL_ADVECTION3D, 2, 4, 1, [0.05]),
                                              (system.KERNEL_DIFFUSION3D, (2, 2, 1), 1, 1, [1.0, 1.0]), (system.KERNEL_DIFFUSION3D, (2, 2, 1), 3, 1, [0.7, 1.3]),
                                              (system.KERNEL_DIFFUSION3D, (2, 1, 1), 5, 1, [1.0, 1.0]), (system.KERNEL_DIFFUSION3D, (2, 1, 1), 7, 1, [0.7, 1.3])])
def test_stored_row_major_routes_agree(ctx, kid, ne, p, vo, kpar):
    """l3k_local_assemble(K) -- the stand-in for assembleLocalSystem's return value (row-major K_e, AssembleLocalSystem.hpp:168-182) --
    forms the matrices in the x-major tiled layout and turns them with the one-pass mirroring transposition kernel on a second
    stream, sub-batch by sub-batch (round 4; the assembly kernel's own row-major stores fill an eighth of each 64-byte line per
    instruction).  The routes agree: default == plain tiled layout + plain transposition (K[i][j], K[j][i] in two summation orders)
    to rounding == the direct store to rounding; the default is symmetric bit for bit like the reference's matrix; several
    sub-batches give the same bits as one."""
    info = system.kernel_info(kid)
    U, F = info["n_unknowns"], info["n_fields"]
    part = system.CubePartition(ne, p, perturb=0.15)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U), kid, kpar, asm_opts=(vo, 0, 0))
    if F:
        mf.set_fields(dev(np.random.default_rng(2).uniform(-1, 1, (F, part.n_local_nodes))))
    K0, _, _ = mf.local_assemble(want_F=False)
    with ctx.tuning(assemble_sub_batch=3):
        K1, _, _ = mf.local_assemble(want_F=False)
    with ctx.tuning(assemble_no_symmetrise=1, assemble_sub_batch=3):
        K2, _, _ = mf.local_assemble(want_F=False)
    with ctx.tuning(assemble_direct_store=1):
        K3, _, _ = mf.local_assemble(want_F=False)
    torch.cuda.synchronize()
    scale = float(K3.abs().amax())
    # (the default route: tiled + transposition from order 4, the direct store below -- small matrices)
    assert torch.equal(K0, K1 if p >= 4 else K3)
    assert torch.equal(K1, K1.transpose(1, 2)) and torch.equal(K3, K3.transpose(1, 2))
    assert float((K1 - K2).abs().amax()) < 1e-13 * scale
    assert float((K2 - K2.transpose(1, 2)).abs().amax()) < 1e-13 * scale
    assert float((K1 - K3).abs().amax()) < 1e-13 * scale
    # a sub-range with an offset, and the checksum beside the stored matrices
    with ctx.tuning(assemble_sub_batch=2):
        Ks, _, cs = mf.local_assemble(1, part.n_elems - 1, want_F=False, want_checksum=True)
    _, _, cs_stream = mf.local_assemble(1, part.n_elems - 1, want_K=False, want_F=False, want_checksum=True)
    assert torch.equal(Ks, K1[1:])
    np.testing.assert_allclose(cs.cpu().numpy(), cs_stream.cpu().numpy(), rtol=1e-12)  # (one atomic add per workgroup: order varies)
