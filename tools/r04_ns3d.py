"""The reference's micro-benchmark workloads through the kernel API on the device: NS3D (benchmarks/Kernels.hpp:3-65,
U = 7, E = 8, F = 7, nq = 2p) as a run-time plugin at p = 2 and 4 -- LocalAssembly (element matrices/s, streaming checksum mode) and the
operator apply (dof/s) with the flop models the reference prints (benchmarks/LocalAssemblyBenchmarks.cpp:71-75,
LocalOperatorEvaluationBenchmarks.cpp:39-42), and the route each shape takes.  One JSON line per order.

    python tools/r04_ns3d.py [--ne 24] > profiles/r04_ns3d.jsonl
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from l3ster_amd import plugin, system  # noqa: E402

KID, U, E, F = 1013, 7, 8, 7


def median_ms(fn, steps):
    for _ in range(2):
        fn()
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    for i in range(steps):
        e0[i].record()
        fn()
        e1[i].record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in zip(e0, e1)]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ne", type=int, default=24)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    src = open(os.path.join(ROOT, "tests", "kernels", "ns3d.hpp")).read()
    plugin.compile_kernel("NS3D", src, KID, shapes=[(2, 4, 1), (4, 8, 1), (6, 12, 1)])
    torch.cuda.set_device(0)
    ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
    ctx.set_tuning(generic_below=0)
    for p in (2, 4, 6):
        nq, n_nodes = 2 * p, (p + 1) ** 3
        part = system.CubePartition(a.ne if p < 6 else min(a.ne, 10), p, perturb=0.1)
        mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U, part.dirichlet_mask(U, unknowns=(0, 1, 2))), KID, asm_opts=(1, 1, 0))
        mf.set_fields(system.synthetic_vector_torch(part.node_grid_id, F, "cuda", seed=7).view(-1, F).t().contiguous())
        X = system.synthetic_vector_torch(part.node_grid_id, U, "cuda")
        Y = torch.zeros_like(X)
        n_qp, rows = nq ** 3, n_nodes * F
        # the reference's flop models, per quadrature point (it counts the local-element algorithm's work)
        asm_flops_qp = n_nodes * 18 + F * n_nodes * 8 + rows * E * 7 + (rows + 1) * (rows + 1) // 2 * (2 * E + 1)
        eval_flops_qp = n_nodes * 18 + F * n_nodes * 8 + E * n_nodes * U * 7 + E * (n_nodes * U * 4 + 2)
        out = {"kernel": "NS3D (benchmarks/Kernels.hpp:3-65) as a plugin", "order": p, "nq": nq, "U": U, "E": E, "F": F,
               "elements": part.n_elems, "dofs": part.n_global_nodes * U, "apply_route": mf.route()}
        ms = median_ms(lambda: mf.apply_elems(2, X, None, Y, None, 1.0, 0.0), a.steps)
        out.update(apply_ms=ms, apply_dof_per_s=out["dofs"] / (ms * 1e-3), apply_ns_per_element=ms * 1e6 / part.n_elems,
                   apply_reference_model_gflops=part.n_elems * n_qp * eval_flops_qp / (ms * 1e-3) / 1e9)
        if p == 2:  # the other route of this shape
            with ctx.tuning(generic_below=10 ** 9):
                out["apply_generic_route"] = mf.route()
                out["apply_generic_ms"] = median_ms(lambda: mf.apply_elems(2, X, None, Y, None, 1.0, 0.0), a.steps)
        batch = min(part.n_elems, {2: 4096, 4: 512, 6: 128}[p])
        ms = median_ms(lambda: mf.local_assemble(0, batch, want_K=False, want_F=False, want_checksum=True), a.steps)
        out.update(assembly_batch=batch, assembly_ms_per_batch=ms, assembly_matrices_per_s=batch / (ms * 1e-3),
                   assembly_reference_model_tflops=batch * n_qp * asm_flops_qp / (ms * 1e-3) / 1e12,
                   assembly_matrix_bytes=(n_nodes * U) ** 2 * 8)
        d, r = mf.diag_rhs(None)
        ms = median_ms(lambda: mf.diag_rhs(None, diag=d, rhs=r), 3)
        out.update(diag_rhs_ms=ms)
        print(json.dumps(out), flush=True)
        del mf, X, Y


if __name__ == "__main__":
    main()
