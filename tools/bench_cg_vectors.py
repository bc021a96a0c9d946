"""HBM rate of the PCG's two fused vector kernels (l3k_cg_update_z: reads z, Ap, minv, writes z + two dot products;
l3k_cg_update_px: reads z, p, x, writes p, x) on vectors of config 5's size.     python tools/bench_cg_vectors.py [--n 67898372]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from l3ster_amd import capi, system  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=67898372)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
torch.cuda.set_device(0)
ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
lib = capi.load()
n = a.n
vp = lambda t: t.data_ptr()
z, q, minv, p, x = (torch.rand(n, dtype=torch.float64, device="cuda") + 0.5 for _ in range(5))
s = torch.tensor([1.0, 3.0, 1.0, 1.0, 0, 0, 0, 0], dtype=torch.float64, device="cuda")


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps


def uz():
    s[:4] = torch.tensor([1.0, 3.0, 1.0, 1.0], dtype=torch.float64, device="cuda")
    capi.check(lib.l3k_cg_update_z(ctx._h, vp(z), vp(q), vp(minv), n, vp(s)))


def upx():
    capi.check(lib.l3k_cg_update_px(ctx._h, vp(p), vp(x), vp(z), n, vp(s)))


t_z, t_px = timed(uz), timed(upx)
print(json.dumps({"n": n, "update_z_ms": t_z, "update_z_GBps": 4 * 8 * n / t_z / 1e6, "update_px_ms": t_px, "update_px_GBps": 5 * 8 * n / t_px / 1e6}))
