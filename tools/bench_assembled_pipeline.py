"""The assembled path end to end on the device (SURVEY.md a8 + a20): l3k_local_assemble with the element matrices STORED,
then l3k_assembled_scatter of the batch into the values of a CSR graph -- what assembleGlobalSystem does element by element
(algsys/AssembleGlobalSystem.hpp:20-53 -> algsys/ScatterLocalSystem.hpp:24-54).  Elements/s of each stage and of the pipeline,
with the algorithmic HBM bytes behind them: K_e written once (Nd^2 * 8 B per element) and read once, the CSR values touched by
Nd^2 atomic adds per element.

    python tools/bench_assembled_pipeline.py [--ne 32] [--orders 2,3,4] [--batch-mb 1024]

Prints one JSON line per order.  The CSR graph is built on the device (unique (row node, column node) pairs), the way the
caller's Tpetra graph would hold it: rows of one node have the same columns, the dofs of a node are neighbours in a row."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from l3ster_amd import system  # noqa: E402


def csr_graph_device(part, U):
    """row_ptr (int64), col_ind (int32) of the dof graph of all elements, built with torch on the GPU from node pairs."""
    en = torch.as_tensor(part.elem_nodes.astype(np.int64), device="cuda")
    n_nodes = part.n_local_nodes
    N = en.shape[1]
    chunks = []
    step = max(1, (1 << 26) // (N * N))  # <= 64 M pairs per chunk
    for i in range(0, en.shape[0], step):
        e = en[i:i + step]
        key = (e[:, :, None] * n_nodes + e[:, None, :]).reshape(-1)
        chunks.append(torch.unique(key))
    key = torch.unique(torch.cat(chunks))
    rn, cn = key // n_nodes, key % n_nodes
    cnt = torch.bincount(rn, minlength=n_nodes)  # column nodes per row node
    node_ptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device="cuda")
    node_ptr[1:] = torch.cumsum(cnt, 0)
    # dof rows: node r -> rows r*U + u, each with the columns (cn*U + u') of its node pairs, ascending
    cols_node = (cn[:, None] * U + torch.arange(U, device="cuda")[None, :]).reshape(-1)  # per node pair, U columns
    row_len = (cnt * U).repeat_interleave(U)
    row_ptr = torch.zeros(n_nodes * U + 1, dtype=torch.int64, device="cuda")
    row_ptr[1:] = torch.cumsum(row_len, 0)
    # the columns of row (r, u) are those of node r for every u: node r's block of cols_node goes to its U rows
    seg = torch.repeat_interleave(torch.arange(n_nodes, device="cuda"), cnt * U)  # node of each entry of cols_node
    col_ind = torch.empty(int(row_ptr[-1].item()), dtype=torch.int32, device="cuda")
    start_node = node_ptr[:-1] * U  # offset of node r's block in cols_node
    ent_off = torch.arange(cols_node.numel(), device="cuda") - start_node[seg]
    for u in range(U):
        col_ind[row_ptr[seg * U + u] + ent_off] = cols_node.to(torch.int32)
    return row_ptr, col_ind


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ne", type=int, default=32)
    ap.add_argument("--orders", default="2,3,4")
    ap.add_argument("--batch-mb", type=int, default=2048, help="size of the stored batch of element matrices")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--per-entry", action="store_true", help="the round-2 scatter (a binary search per entry) as the baseline")
    a = ap.parse_args()
    if a.per_entry:
        os.environ["L3K_SCATTER_PER_ENTRY"] = "1"
    torch.cuda.set_device(0)
    ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
    U, kid = 4, system.KERNEL_DIFFUSION3D
    for p in [int(s) for s in a.orders.split(",")]:
        part = system.CubePartition(a.ne, p, perturb=0.1)
        mesh = system.DeviceMesh(ctx, part, U)
        mf = system.MatrixFreeSystem(mesh, kid, [1.0, 1.0])
        Nd = (p + 1) ** 3 * U
        kbytes = Nd * Nd * 8
        batch = max(1, min(part.n_elems, (a.batch_mb << 20) // kbytes))
        row_ptr, col_ind = csr_graph_device(part, U)
        values = torch.zeros(col_ind.numel(), dtype=torch.float64, device="cuda")
        rhs = torch.zeros((1, part.n_local_nodes * U), dtype=torch.float64, device="cuda")
        n_sweep = min(part.n_elems, 8 * batch)

        def sweep(assemble, scatter):
            K = Fe = None
            for first in range(0, n_sweep, batch):
                cnt = min(batch, n_sweep - first)
                if assemble:
                    K, Fe, _ = mf.local_assemble(first, cnt)
                if scatter:
                    if K is None or K.shape[0] != cnt:
                        K, Fe, _ = mf.local_assemble(first, cnt)
                    mf.assembled_scatter(K, Fe, row_ptr, col_ind, values, rhs, first=first)
            return K, Fe

        def timed(assemble, scatter):
            K, Fe = sweep(True, False) if (scatter and not assemble) else (None, None)  # a resident batch for the scatter alone
            t = []
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                if scatter and not assemble:
                    for first in range(0, n_sweep, batch):
                        cnt = min(batch, n_sweep - first)
                        mf.assembled_scatter(K[:cnt], Fe[:cnt], row_ptr, col_ind, values, rhs, first=first)
                else:
                    sweep(assemble, scatter)
                e1.record()
                torch.cuda.synchronize()
                t.append(e0.elapsed_time(e1) * 1e-3)
            return float(np.median(t))

        t_asm, t_sc, t_both = timed(True, False), timed(False, True), timed(True, True)
        # the same sweep as one call per sweep: l3k_assemble_global (assembly and scatter of consecutive sub-batches on two streams)
        t = []
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            mf.assemble_global(row_ptr, col_ind, values, rhs, first=0, count=n_sweep, workspace_bytes=a.batch_mb << 20)
            e1.record()
            torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1) * 1e-3)
        t_glob = float(np.median(t))
        # parity of the pipeline on the way: the assembled operator applied to x equals the matrix-free apply (no Dirichlet dofs)
        values.zero_()
        rhs.zero_()
        sweep_all = n_sweep == part.n_elems
        err = None
        if sweep_all:
            mf.assemble_global(row_ptr, col_ind, values, rhs, workspace_bytes=a.batch_mb << 20)
            x = system.synthetic_vector_torch(part.node_grid_id, U, "cuda")
            A = torch.sparse_csr_tensor(row_ptr, col_ind.to(torch.int64), values, size=(x.shape[1], x.shape[1]))
            y_asm = (A @ x[0].unsqueeze(1)).squeeze(1)
            y_mf = torch.zeros_like(x)
            mf.apply(x, y_mf, 1.0, 0.0)
            err = float((y_asm - y_mf[0]).norm() / y_mf[0].norm())
        out = {"scatter": "search per entry (round 2)" if a.per_entry else "search per node pair",
               "workload": f"Diffusion3D assembled pipeline, hex mesh {a.ne}^3, order {p}: {n_sweep} elements in batches of {batch}",
               "Nd": Nd, "K_e_bytes": kbytes, "csr_nnz": int(col_ind.numel()),
               "local_assemble_stored": {"elements_per_s": n_sweep / t_asm, "write_GBps": n_sweep * kbytes / t_asm / 1e9},
               "assembled_scatter": {"elements_per_s": n_sweep / t_sc, "read_GBps": n_sweep * kbytes / t_sc / 1e9,
                                     "atomic_adds_per_s": n_sweep * Nd * Nd / t_sc},
               "assemble_global_two_streams": {"elements_per_s": n_sweep / t_glob},
               "pipeline": {"elements_per_s": n_sweep / t_both, "GBps_K_written_plus_read": 2 * n_sweep * kbytes / t_both / 1e9},
               "assembled_vs_matrix_free_rel_l2": err}
        print(json.dumps(out), flush=True)
        del mf, mesh, values, rhs, row_ptr, col_ind


if __name__ == "__main__":
    main()
