"""Kernel micro-benchmark: times the element kernel alone (HIP events on the launch stream) for one shape, with the
ablation switches of L3K_DEBUG_FLAGS (1 no scatter, 2 no gather loads, 4 no QP stage, 8 no sweeps, 16 plain stores
instead of atomics, 128 every other shell round of the scatter dropped: half the atomics).  Usage: python tools/kbench.py --order 6 --ne 32 [--flags 0,1,2,...]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_one(order, ne, steps, perturb):
    import numpy as np
    import torch
    from l3ster_amd import system
    torch.cuda.set_device(0)
    ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
    U = 4
    part = system.CubePartition(ne, order, perturb=perturb)
    mesh = system.DeviceMesh(ctx, part, U, part.dirichlet_mask(U))
    mf = system.MatrixFreeSystem(mesh, system.KERNEL_DIFFUSION3D, [1.0, 1.0])
    X = system.synthetic_vector_torch(part.node_grid_id, U, "cuda")
    Y = torch.zeros_like(X)
    for _ in range(3):
        mf.apply_elems(2, X, None, Y, None, 1.0, 0.0)
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    for i in range(steps):
        e0[i].record()
        mf.apply_elems(2, X, None, Y, None, 1.0, 0.0)
        e1[i].record()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in zip(e0, e1)])
    dofs = part.n_global_nodes * U
    if os.environ.get("L3K_STAMPS"):
        stage_timeline()
    print(f"flags={os.environ.get('L3K_DEBUG_FLAGS', '0'):>3} p={order} ne={ne} elems={part.n_elems} dofs={dofs} "
          f"ms(min/med)={ms.min():.3f}/{np.median(ms):.3f}  DOF/s={dofs / np.median(ms) * 1e3:.3e}  "
          f"ns/elem={np.median(ms) * 1e6 / part.n_elems:.1f}", flush=True)


STAGES = ["gather+S1(z interp)", "S2 (y interp)", "S3/S4 (x interp, d/dxi)", "S5 (d/deta)", "S6 (d/dzeta)", "QP stage",
          "S8/S9 (C^T eta, zeta)", "combine + C^T xi + I^T x", "I^T y", "I^T z + staging", "ids to LDS", "scatter issue"]


def stage_timeline():
    """Cycle counters of workgroup 0 at its stage boundaries (last launch), L3K_ABLATION build with L3K_STAMPS=1."""
    import ctypes as C
    import numpy as np
    from l3ster_amd import capi
    lib = capi.load()
    buf = np.zeros(256 * 16 + 2 * 4096, dtype=np.int64)
    fn = lib.l3k_debug_stamps
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_int64), C.c_int]
    if fn(buf.ctypes.data_as(C.POINTER(C.c_int64)), buf.size) != 0:
        print("no stamps (needs the ablation library and L3K_STAMPS=1)")
        return
    life = buf[256 * 16:].reshape(4096, 2)
    life = life[(life[:, 0] != 0) & (life[:, 1] != 0)]
    if os.environ.get("L3K_STAMPS_DUMP"):
        np.save(os.environ["L3K_STAMPS_DUMP"], buf)
    if len(life):
        # (the clock is per CU / shader engine: only end - start of one workgroup is meaningful, not differences between them)
        d = life[:, 1] - life[:, 0]
        print(f"workgroup lifetimes (cycles), {len(d)} workgroups: min {d.min()} p5 {int(np.percentile(d, 5))} median "
              f"{int(np.median(d))} p95 {int(np.percentile(d, 95))} max {d.max()}; mean / max = {d.mean() / d.max():.3f}")
        for x in range(8):
            dx = d[x::8]
            print(f"  XCD {x}: min {dx.min()} median {int(np.median(dx))} max {dx.max()}  mean / max = {dx.mean() / dx.max():.3f}")
    t = buf[:256 * 16].reshape(256, 16)
    n = 1  # (the buffer is not cleared between launches: the last launch's iterations are the increasing prefix)
    while n < 256 and t[n, 0] > t[n - 1, 0]:
        n += 1
    if n < 4:
        print("stamps: too few iterations recorded")
        return
    t = t[1:n - 1]  # steady state
    d = np.diff(t[:, :13], axis=1)
    tot = t[1:, 0] - t[:-1, 0]
    print(f"stage timeline of workgroup 0, {len(t)} elements, cycles per element: median total {np.median(tot):.0f}")
    for name, col in zip(STAGES, d.T):
        print(f"  {name:28s} median {np.median(col):8.0f}  mean {col.mean():8.0f}  ({100 * col.mean() / tot.mean():4.1f} %)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--order", type=int, default=6)
    ap.add_argument("--ne", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--perturb", type=float, default=0.1)
    ap.add_argument("--flags", default="0")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        run_one(a.order, a.ne, a.steps, a.perturb)
    else:
        for f in a.flags.split(","):
            env = dict(os.environ, L3K_DEBUG_FLAGS=f)
            if f != "0":  # ablation switches live in the L3K_ABLATION=1 build only
                lib = os.path.join(ROOT, "l3ster_amd", "lib", "libl3k_ablation.so")
                if not os.path.exists(lib):
                    raise SystemExit("build the ablation library first: L3K_ABLATION=1 python -m l3ster_amd.build")
                env["L3K_LIB"] = lib
            subprocess.run([sys.executable, __file__, "--child", "--order", str(a.order), "--ne", str(a.ne), "--steps",
                            str(a.steps), "--perturb", str(a.perturb)], env=env, check=True)
