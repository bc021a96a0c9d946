"""Occupancy proxy for the order-6 element kernel (round 3).

Question: how does the one-wave-per-element pencil kernel scale with resident waves per CU BEYOND the 7-8 that the
order-6 Diffusion3D instance can hold (22 KB of LDS, 256 VGPRs per wave)?  The real instance cannot answer it (its LDS
and registers pin it), so this runs the SAME kernel template on a two-unknown kernel of the same per-field cost
(`ProxyU2`: half the fields => half the LDS (11 KB) and roughly half the registers per wave), compiled with
__launch_bounds__(64, MIN_WAVES) for MIN_WAVES = 2, 3, 4 (256 / 168 / 128 VGPRs), at 4 ... 14 waves per CU.

    L3K_FAST_MIN_WAVES=3 python tools/r03_occupancy_proxy.py --waves 4,6,7,8,10,12 [--compile-only]

Prints ns per element and the compiler's register / spill counts of the instance.
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SRC = '''
struct ProxyU2
{
    // (T, q_x) with four first-order rows of the Diffusion3D kind: the same density of A_i entries per unknown
    static constexpr l3k::KernelParams params{.dimension = 3, .n_equations = 4, .n_unknowns = 2};
    double k = 1.;
    template < typename In, typename Out >
    L3K_HD void operator()(const In&, Out& out) const
    {
        auto& [operators, rhs] = out;
        auto& [A0, Ax, Ay, Az] = operators;
        Ax(0, 1) = -k;
        rhs[0]   = 1.;
        A0(1, 1) = -1.;
        Ax(1, 0) = 1.;
        Ay(2, 0) = 1.;
        Az(2, 1) = -1.;
        Az(3, 0) = 1.;
        Ay(3, 1) = 1.;
    }
};'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--waves", default="4,6,7,8")
    ap.add_argument("--ne", type=int, default=48)
    ap.add_argument("--order", type=int, default=6)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--compile-only", action="store_true")
    a = ap.parse_args()
    from l3ster_amd import plugin
    kid = plugin.compile_kernel("ProxyU2", SRC, kernel_id=1900, shapes=[(a.order, a.order + 1, 1)], verbose=True)
    mw = os.environ.get("L3K_FAST_MIN_WAVES", "2")
    if a.compile_only:
        print(f"compiled ProxyU2 order {a.order} with L3K_FAST_MIN_WAVES={mw}")
        return
    import numpy as np
    import torch
    from l3ster_amd import system
    torch.cuda.set_device(0)
    ctx = system.Context(0, torch.cuda.current_stream().cuda_stream)
    U = 2
    part = system.CubePartition(a.ne, a.order, perturb=0.1)
    mesh = system.DeviceMesh(ctx, part, U, part.dirichlet_mask(U))
    mf = system.MatrixFreeSystem(mesh, kid, [1.0])
    X = system.synthetic_vector_torch(part.node_grid_id, U, "cuda")
    Y = torch.zeros_like(X)
    for w in [int(s) for s in a.waves.split(",")]:
        os.environ["L3K_FAST_WAVES_PER_CU"] = str(w)
        for _ in range(3):
            mf.apply_elems(2, X, None, Y, None, 1.0, 0.0)
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
        for i in range(a.steps):
            e0[i].record()
            mf.apply_elems(2, X, None, Y, None, 1.0, 0.0)
            e1[i].record()
        torch.cuda.synchronize()
        ms = np.array([x.elapsed_time(y) for x, y in zip(e0, e1)])
        print(f"ProxyU2 p={a.order} ne={a.ne} min_waves={mw} waves/CU={w:2d}: ms(min/med)={ms.min():.3f}/{np.median(ms):.3f} "
              f"ns/elem={np.median(ms) * 1e6 / part.n_elems:.2f}", flush=True)


if __name__ == "__main__":
    main()
