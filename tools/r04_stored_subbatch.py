import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from l3ster_amd import system, capi
torch.cuda.set_device(0)
ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
p, U, batch = 6, 4, 512
part = system.CubePartition(8, p, perturb=0.1)
mf = system.MatrixFreeSystem(system.DeviceMesh(ctx, part, U), system.KERNEL_DIFFUSION3D, [1.0, 1.0])
Nd = (p + 1) ** 3 * U
K = torch.empty((batch, Nd, Nd), dtype=torch.float64, device="cuda")
def timeit(fn, steps=3):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
ms = timeit(lambda: capi.check(capi.load().l3k_local_assemble_tiled(mf._h, 0, batch, C.c_void_p(K.data_ptr()))))
print(f"tiled alone batch {batch}: {ms:.3f} ms = {batch/ms:.1f} k/s")
for sb in (16, 32, 64, 128, 256, 512):
    for nosym in (1, 0):
        with ctx.tuning(assemble_sub_batch=sb, assemble_no_symmetrise=nosym):
            ms = timeit(lambda: capi.check(capi.load().l3k_local_assemble(mf._h, 0, batch, C.c_void_p(K.data_ptr()), None, None)))
        print(f"sub-batch {sb:4d} nosym={nosym}: {ms:.3f} ms = {batch/ms:.1f} k/s", flush=True)
