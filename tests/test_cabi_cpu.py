"""CPU-side checks of the product library: it loads, exports every symbol include/l3k.h declares, its host logic
(tables, mesh generator / partition, exchange plan) is right, and the device entry points fail loudly without a GPU.
No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from l3ster_amd import capi, system

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "l3k.h")).read()
    declared = set(re.findall(r"\b(l3k_[a-z0-9_]+)\s*\(", header))
    lib = capi.load()
    assert declared == set(capi.SIGNATURES), declared ^ set(capi.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.l3k_version() == 101


def test_hand_written_dpp_instructions_have_no_hazards():
    """device/assemble.hpp issues v_fmac_f64_dpp ... row_newbcast through inline asm, which the compiler's hazard recognizer does
    not see: the ISA of the built library is scanned for the two DPP hazards of the ISA manual (a VALU write of the DPP source
    within 2 wait states, a VALU write of EXEC within 5) -- tools/check_dpp_hazards.py."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_dpp_hazards", os.path.join(root, "tools", "check_dpp_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, bad = mod.check(os.path.join(root, "l3ster_amd", "lib", "libl3k.so"))
    assert n > 1000, "the DPP kernels are gone from the library?"
    assert not bad, bad[:5]


def test_isa_scan_follows_branches_and_refuses_what_it_cannot_read(tmp_path, monkeypatch):
    """the scan (a) used to pass when it found no gfx950 code object at all (a compressed bundle, another layout),
    (b) did not look across loop back-edges: a VALU write of a DPP source at a loop's tail is a hazard for a DPP read at the head."""
    from l3ster_amd import isa_check
    junk = tmp_path / "not_a_library.so"
    junk.write_bytes(b"\x7fELF" + b"CCOB" + bytes(200))
    with pytest.raises(isa_check.IsaScanError, match="compressed offload bundle"):
        isa_check.scan(str(junk))
    junk.write_bytes(b"\x7fELF" + bytes(200))
    with pytest.raises(isa_check.IsaScanError, match="no gfx950 code object"):
        isa_check.check_dpp_hazards(str(junk))
    # a loop: head at 0x100 reads v[4:5] through DPP, the tail writes v[4:5] and branches back (simm16 = -7 dwords from 0x11c)
    listing = """
0000000000000100 <loop_kernel>:
\tv_fmac_f64_dpp v[0:1], v[4:5], v[2:3] row_newbcast:3 row_mask:0xf bank_mask:0xf // 000000000100: 00000000 00000000
\tv_add_f64 v[6:7], v[0:1], v[2:3]                          // 000000000108: 00000000 00000000
\ts_nop 0                                                   // 000000000110: BF800000
\tv_mov_b32_e32 v4, v8                                      // 000000000114: 7E080308
\ts_cbranch_scc1 65529                                      // 000000000118: BF85FFF9 <loop_kernel>
\ts_endpgm                                                  // 00000000011C: BF810000
"""
    monkeypatch.setattr(isa_check, "code_objects", lambda path: iter([b"x"]))
    monkeypatch.setattr(isa_check, "_disassemble", lambda co: listing)
    junk.write_bytes(b"anything")
    r = isa_check.scan(str(junk))
    assert r["n_dpp"] == 1 and len(r["hazards"]) == 1 and "across the branch" in r["hazards"][0][2], r
    # the same loop with two wait states between the write and the branch is clean
    monkeypatch.setattr(isa_check, "_disassemble", lambda co: listing.replace("\tv_mov_b32_e32 v4, v8 ", "\tv_mov_b32_e32 v4, v8 ").replace(
        "\ts_cbranch_scc1 65529                                      // 000000000118: BF85FFF9 <loop_kernel>",
        "\ts_nop 1                                                   // 000000000118: BF800001\n\ts_cbranch_scc1 65528                                      // 00000000011C: BF85FFF8 <loop_kernel>"))
    assert isa_check.scan(str(junk))["hazards"] == []


def test_tables_match_oracle_and_golden(golden):
    g = golden("tables")
    for p in range(1, 9):
        np.testing.assert_allclose(system.gll_nodes(p + 1), g[f"gll_{p}"], atol=1e-15, rtol=0)
        for nq in sorted({p + 1, 2 * p + 1}):
            x, w = system.gl_rule(nq)
            np.testing.assert_allclose(x, g[f"qx_{nq}"], atol=2e-16, rtol=0)
            np.testing.assert_allclose(w, g[f"qw_{nq}"], atol=1e-15, rtol=0)
            I, D = system.basis_1d(p, nq)
            np.testing.assert_allclose(I, g[f"I_{p}_{nq}"], atol=2e-15, rtol=0)
            np.testing.assert_allclose(D, g[f"D_{p}_{nq}"], atol=1e-13, rtol=0)
            # collocation identity behind the device algorithm: D = I * C for nq >= p+1
            np.testing.assert_allclose(I @ system.colloc_deriv(nq), D, atol=1e-12, rtol=0)
    assert system.n_qps1d(6) == 7 and system.n_qps1d(3, 2) == 7 and system.n_qps1d(4, 1, 1) == 8


def test_kernel_registry_matches_oracle():
    for kid in (system.KERNEL_DIFFUSION3D, system.KERNEL_DIFFUSION3D_VAR, system.KERNEL_ADVDIFF3D):
        a, b = system.kernel_info(kid), O.kernel_params(kid)
        assert (a["dimension"], a["n_equations"], a["n_unknowns"], a["n_fields"]) == (b["dim"], b["E"], b["U"], b["F"])
    shapes = system.instances()
    assert (0, 6, 7, 1) in shapes and (0, 4, 5, 1) in shapes and (4, 4, 5, 1) in shapes
    with pytest.raises(system.L3KError, match="unknown kernel"):
        system.kernel_info(99)


@pytest.mark.parametrize("ne,p", [(3, 1), (2, 2), (4, 3), ((3, 2, 4), 4), (2, 6)])
def test_single_part_mesh_conventions(ne, p):
    m = system.CubePartition(ne, p, perturb=0.1)
    ne3 = (ne,) * 3 if np.isscalar(ne) else ne
    n = p + 1
    assert m.n_elems == np.prod(ne3) == m.n_interior_elems and m.n_ghost_nodes == 0
    assert m.n_owned_nodes == np.prod([p * e + 1 for e in ne3]) == m.n_global_nodes
    assert np.array_equal(np.unique(m.elem_nodes), np.arange(m.n_owned_nodes))  # a numbering of all nodes
    # element-internal nodes are numbered after every non-internal node, contiguous per element, lexicographic
    # (mesh/LocalMeshView.hpp:425-458)
    idx = np.arange(n ** 3)
    ix, iy, iz = idx % n, (idx // n) % n, idx // (n * n)
    internal = (ix > 0) & (ix < p) & (iy > 0) & (iy < p) & (iz > 0) & (iz < p)
    n_int = (p - 1) ** 3
    if n_int:
        first_internal = m.n_owned_nodes - m.n_elems * n_int
        ids = m.elem_nodes[:, internal].astype(np.int64)
        assert ids.min() == first_internal
        assert np.all(np.diff(ids, axis=1) == 1)
        assert m.elem_nodes[:, ~internal].max() < first_internal
    # geometry: element nodes (tri-linear map of the 8 vertices) agree between neighbouring elements
    Nx = p * ne3[0] + 1
    Ny = p * ne3[1] + 1
    gx, gy, gz = m.node_grid_id % Nx, (m.node_grid_id // Nx) % Ny, m.node_grid_id // (Nx * Ny)
    sides = m.node_boundary
    assert np.array_equal((sides & 16) != 0, gx == 0) and np.array_equal((sides & 32) != 0, gx == Nx - 1)
    assert np.array_equal((sides & 1) != 0, gz == 0) and np.array_equal((sides & 4) != 0, gy == 0)
    gll = system.gll_nodes(n)
    pos = {}
    for e in range(m.n_elems):
        for i in (0, n - 1, n * n - 1, n ** 3 - 1, (n ** 3) // 2):
            xyz = O.map_to_physical(3, m.elem_verts[e], [gll[i % n], gll[(i // n) % n], gll[i // (n * n)]])
            node = m.elem_nodes[e, i]
            if node in pos:
                np.testing.assert_allclose(xyz, pos[node], atol=1e-14)
            pos[node] = xyz


@pytest.mark.parametrize("ne,p,parts", [(4, 3, (2, 2, 2)), ((4, 2, 3), 2, (2, 1, 3)), (3, 1, (3, 3, 3)), (4, 4, (1, 2, 1))])
def test_partition_ownership_and_exchange_plan(ne, p, parts):
    """util/SegmentedOwnership.hpp:11-45 (contiguous owned ranges, shared sorted by global id),
    comm/ImportExport.hpp:29-72 (who shares my owned indices / who owns my shared ones)."""
    check_partition(ne, p, parts)


def test_partition_invariants_random_shapes():
    """The same invariants on randomly drawn meshes, uneven block splits included (hypothesis)."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=30, deadline=None)
    @given(st.tuples(st.integers(1, 5), st.integers(1, 5), st.integers(1, 4)), st.integers(1, 3), st.data())
    def run(ne, p, data):
        parts = tuple(data.draw(st.integers(1, min(n, 3))) for n in ne)
        check_partition(ne, p, parts)

    run()


def check_partition(ne, p, parts):
    nparts = int(np.prod(parts))
    pm = [system.CubePartition(ne, p, parts, r) for r in range(nparts)]
    whole = system.CubePartition(ne, p)
    assert sum(q.n_owned_nodes for q in pm) == whole.n_global_nodes
    assert sum(q.n_elems for q in pm) == whole.n_elems
    base = 0
    owner_of = {}
    for r, q in enumerate(pm):
        assert q.global_node_base == base
        base += q.n_owned_nodes
        for gid in q.node_grid_id[:q.n_owned_nodes]:
            assert gid not in owner_of  # every node has exactly one owner
            owner_of[int(gid)] = r
        # interior elements touch owned nodes only, border elements touch at least one ghost
        # (algsys/MatrixFreeSystem.hpp:969-981)
        assert q.elem_nodes[:q.n_interior_elems].max(initial=0) < max(q.n_owned_nodes, 1)
        if q.n_elems > q.n_interior_elems:
            assert np.all(q.elem_nodes[q.n_interior_elems:].max(axis=1) >= q.n_owned_nodes)
    assert len(owner_of) == whole.n_global_nodes
    for r, q in enumerate(pm):
        ghost_ids = q.node_grid_id[q.n_owned_nodes:]
        assert all(owner_of[int(g)] < r for g in ghost_ids)  # lowest part touching a node owns it
        covered = 0
        for i, nb in enumerate(q.nbr_rank):
            g0, g1 = q.ghost_ranges[i]
            covered += g1 - g0
            o = pm[nb]
            j = o.nbr_rank.index(r)
            if g1 > g0:
                assert nb < r and np.all(np.diff(o.send_nodes[j]) > 0)
                assert np.array_equal(o.node_grid_id[o.send_nodes[j]], ghost_ids[g0:g1])
            else:
                og0, og1 = o.ghost_ranges[j]
                assert nb > r and og1 - og0 == len(q.send_nodes[i]) > 0
        assert covered == q.n_ghost_nodes


def test_partition_independent_synthetic_vector():
    import torch
    whole = system.CubePartition(3, 2)
    ref = dict(zip(whole.node_grid_id.tolist(), whole.synthetic_vector(4).reshape(-1, 4)))
    for r in range(4):
        q = system.CubePartition(3, 2, (2, 2, 1), r)
        x = q.synthetic_vector(4).reshape(-1, 4)
        for gid, row in zip(q.node_grid_id.tolist(), x):
            assert np.array_equal(row, ref[gid])
        xt = system.synthetic_vector_torch(q.node_grid_id, 4, "cpu").numpy().reshape(-1, 4)
        assert np.array_equal(xt, x)
    assert abs(whole.synthetic_vector(4).mean()) < 0.1


def test_device_entry_points_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(system.L3KError, match="no HIP device|no CPU fallback"):
        system.Context(0)


def test_bad_arguments_are_reported_not_crashed():
    with pytest.raises(system.L3KError):
        system.CubePartition(4, 3, (5, 1, 1), 0)  # more parts than elements
    with pytest.raises(system.L3KError):
        system.CubePartition(4, 3, (2, 1, 1), 2)  # rank outside the partition
    with pytest.raises(system.L3KError):
        system.gll_nodes(1)
