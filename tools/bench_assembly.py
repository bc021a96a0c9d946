"""Element-matrices/s of LocalAssembly (BASELINE.json config 3: Diffusion3D, hex, order 6): the sum-factorised assembly
kernel by default, the dense K_e = B^T W B product on the FP64 matrix cores with L3K_ASSEMBLE_DENSE=1.  Streaming mode: batches of elements, K_e reduced to a checksum on the device (the 64^3 mesh's matrices
would be 3.9 TB); optionally stored.  Prints one JSON line.
    python tools/bench_assembly.py [--order 6] [--batch 64] [--steps 5] [--store]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from l3ster_amd import system  # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 78.6  # AMD spec, vector = matrix; measured on the bench box 49 (MFMA) / 57.5 (vector), tools/fp64_peak.hip

ap = argparse.ArgumentParser()
ap.add_argument("--order", type=int, default=6)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--store", action="store_true")
a = ap.parse_args()
torch.cuda.set_device(0)
ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
p, U, E = a.order, 4, 7
ne = 4
while ne ** 3 < a.batch:
    ne += 1
part = system.CubePartition(ne, p, perturb=0.1)  # at least `batch` elements
mesh = system.DeviceMesh(ctx, part, U)
mf = system.MatrixFreeSystem(mesh, system.KERNEL_DIFFUSION3D, [1.0, 1.0])
mf.local_assemble(0, a.batch, want_K=a.store, want_F=False, want_checksum=True)  # warm-up
torch.cuda.synchronize()
e0 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
e1 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
for i in range(a.steps):
    e0[i].record()
    mf.local_assemble(0, a.batch, want_K=a.store, want_F=False, want_checksum=True)
    e1[i].record()
torch.cuda.synchronize()
ms = float(np.median([x.elapsed_time(y) for x, y in zip(e0, e1)]))
nd, kd = (p + 1) ** 3 * U, (p + 1) ** 3 * E
flops_sym = 1.0 * kd * nd * (nd + 1)  # 2*kd*nd^2/2 (+diagonal): the symmetric half the reference computes (rankUpdate)
rate = a.batch / (ms * 1e-3)
dense = os.environ.get("L3K_ASSEMBLE_DENSE") is not None
print(json.dumps({"metric": "element-matrices/s for assembled path (LocalAssembly, Diffusion3D, hex p=%d)" % p, "value": rate,
                  "algorithm": "dense (W Z)^T Z on v_mfma_f64_16x16x4" if dense else "sum-factorised (the TFLOP/s below are dense-equivalent, not executed)",
                  "unit": "element matrices/s", "ms_per_batch": ms, "batch": a.batch, "stored": a.store, "dtype": "f64",
                  "roofline": {"bound": "mfma", "achieved": rate * flops_sym / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": rate * flops_sym / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                               "flops_per_element_symmetric": flops_sym}}))
