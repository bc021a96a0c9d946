"""Stored LocalAssembly in the reference's layout (row-major K_e, AssembleLocalSystem.hpp:168-182) -- VERDICT r3 item 7: the three
routes of l3k_local_assemble(K) at orders 6 / 4 / 2: the assembly kernel's direct row-major store (round 3: 8-byte stores at a 32-byte
stride, 3.9 x write traffic), tiled + transposition kernel on a second stream (no symmetrisation: K symmetric to rounding), and the
default: the x-major tiled layout + the one-pass transposition that reads the lower triangle and writes every entry twice (bitwise
symmetric like the reference's).  One JSON line per order.

    python tools/r04_stored_assembly.py [--batch 256] > profiles/r04_stored_assembly.jsonl
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from l3ster_amd import system  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--orders", default="6,4,2")
ap.add_argument("--routes", default="direct_store,tiled_transposed,x_tiled_one_pass_symmetric", help="(profiles: one tiled route per run keeps the per-kernel means apart)")
a = ap.parse_args()
torch.cuda.set_device(0)
ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
U = 4
for p in [int(s) for s in a.orders.split(",")]:
    batch = a.batch * {6: 1, 4: 8, 2: 64}.get(p, 1)
    ne = 2
    while ne ** 3 < batch:
        ne += 1
    part = system.CubePartition(ne, p, perturb=0.1)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U), system.KERNEL_DIFFUSION3D, [1.0, 1.0])
    Nd = (p + 1) ** 3 * U
    K = torch.empty((batch, Nd, Nd), dtype=torch.float64, device="cuda")
    out = {"order": p, "batch": batch, "matrix_bytes": Nd * Nd * 8}
    ref = None
    for name, tune in (("direct_store", dict(assemble_direct_store=1)), ("tiled_transposed", dict(assemble_no_symmetrise=1)),
                       ("x_tiled_one_pass_symmetric", dict())):
        if name not in a.routes.split(","):
            continue
        with ctx.tuning(**tune):
            check = lambda rc: None
            from l3ster_amd import capi
            import ctypes as C
            call = lambda: capi.check(capi.load().l3k_local_assemble(mf._h, 0, batch, C.c_void_p(K.data_ptr()), None, None))
            call()
            torch.cuda.synchronize()
            e0 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
            e1 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
            for i in range(a.steps):
                e0[i].record()
                call()
                e1[i].record()
            torch.cuda.synchronize()
            ms = float(np.median([x.elapsed_time(y) for x, y in zip(e0, e1)]))
        sym = bool(torch.equal(K[:8], K[:8].transpose(1, 2)))
        asym = float((K[:8] - K[:8].transpose(1, 2)).abs().amax() / K[:8].abs().amax())
        if ref is None:
            ref = K[:8].clone()
        out[name] = {"ms_per_batch": ms, "matrices_per_s": batch / (ms * 1e-3), "GB_per_s_of_matrices": batch * Nd * Nd * 8 / (ms * 1e-3) / 1e9,
                     "bitwise_symmetric": sym, "max_asymmetry_rel": asym,
                     "max_diff_vs_direct_rel": float((K[:8] - ref).abs().amax() / ref.abs().amax())}
    print(json.dumps(out), flush=True)
    del K, mf
