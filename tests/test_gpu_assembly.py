"""GPU parity tests of LocalAssembly (K_e via FP64 MFMA, F_e): HIP path through the C ABI vs the oracle's
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
    unknown blocks vanish, sum_i F_e[(i,u)] = rhs_u * volume -- through l3k_local_assemble (MFMA path) and through the
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
    for generic_below in ("0", "1000000"):  # the one-wave kernel and the generic LDS kernel
        import os
        os.environ["L3K_GENERIC_BELOW"] = generic_below
        try:
            X, Y = dev(ones[None, :]), dev(np.zeros((1, K.shape[0])))
            mf.apply(X, Y, 1.0, 0.0)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("L3K_GENERIC_BELOW", None)
        y = Y.cpu().numpy()[0]
        np.testing.assert_allclose(y, K_ref @ ones, atol=1e-12)
        assert abs(y[0::2].sum() - vol) < 1e-11 and np.abs(y[1::2]).max() < 1e-13


def test_config3_full_streaming_sweep_64cubed_order6(ctx):
    """BASELINE.json configs[2] as stated: Diffusion3D, hex mesh 64^3, order 6, LocalAssembly (B^T W B on the FP64 matrix
    cores) over the WHOLE mesh in streaming mode (the 262 144 matrices would be 3.9 TB: each K_e is reduced to its weighted
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
    assert np.all(np.isfinite(cs)) and rate > 2000  # (9 900 element matrices/s measured; generous floor)
    Nd = (p + 1) ** 3 * U
    E = np.zeros((Nd, 7))
    E[np.arange(Nd), np.arange(Nd) % 7] = 1.0
    wgt = 1.0 + (3 * (np.arange(Nd)[:, None] + np.arange(7)[None, :])) % 7
    sample = np.random.default_rng(11).choice(part.n_elems, 32, replace=False)
    for e in sample:
        Y = O.apply_sumfact(system.KERNEL_DIFFUSION3D, p, p + 1, part.elem_verts[e], E, kparams=kpar)
        ref = float((Y * wgt).sum())
        assert abs(cs[e] - ref) < 1e-10 * float((np.abs(Y) * wgt).sum()), (e, cs[e], ref)
