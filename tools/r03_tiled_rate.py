import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from l3ster_amd import system
torch.cuda.set_device(0)
ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
U = 4
for p, ne, batch in ((2, 32, 16384), (3, 32, 4096), (4, 20, 1024), (6, 8, 64), (6, 8, 256)):
    part = system.CubePartition(ne, p, perturb=0.1)
    mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U), system.KERNEL_DIFFUSION3D, [1.0, 1.0])
    for name, fn in (("tiled", lambda: mf.local_assemble_tiled(0, batch)), ("row-major", lambda: mf.local_assemble(0, batch, want_F=False)),
                     ("streaming", lambda: mf.local_assemble(0, batch, want_K=False, want_F=False, want_checksum=True))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        Nd = (p + 1) ** 3 * U
        print(f"p={p} {name:10s} batch {batch}: {batch / ms * 1e3:12.0f} matrices/s  {batch * Nd * Nd * 8 / ms / 1e6:8.1f} GB/s of K", flush=True)
