#!/usr/bin/env python
"""bench.py -- DOF/s of the matrix-free sum-factorised Diffusion3D operator apply (hex, order 6) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--ne 64] [--order 6]

N > 1: one process per GPU.  Under a launcher (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) this process
is one rank; typed as above without a launcher it starts its N ranks itself as child processes and relays rank 0's JSON line.

A "step" is one operator apply Y <- A X (alpha = 1, beta = 0) over the whole mesh.  Workload per GPU (weak scaling):
a 64^3-element block of order-6 hexes on [0,1]^3 (228.3 M global dofs on one GPU; 128^3 = BASELINE.json config 3 on
8 GPUs as 2x2x2 blocks), smoothly perturbed vertices (no affine shortcut), Dirichlet on unknown 0 of all six sides,
kernel = benchmarks/Diffusion3D.hpp:51-79 (U = 4, E = 7, k = s = 1), x ~ U(-1,1) synthetic, resident in HBM.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and (N = 1) `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from l3ster_amd import launch, system  # noqa: E402
from l3ster_amd.distributed import DistributedOperator, HaloPlan, HostStagedTransport, NativeDistributedOperator, NativeHalo  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PARTS = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}
# executed FP64 flops per element of sumfactFastKernel<Diffusion3D> (ISA count: (fma + fmac) * 2 + mul + add, times the
# active lanes per element): order 6: 4 460 lane-flops x 49 lanes
FP64_FLOP_PER_ELEM = {6: 4460 * 49}
# executed flops per element of assembleSumfactKernel at order 6, U = 4 (10 unknown pairs u' <= u), per workgroup and iteration
# stage 1: 7 pairs x 49 x 112 FMAs, stage 2: 441 per row, stage 3 (factorised, second step with the even-odd decomposition:
# 392 + 7 * 124 = 1 260 flops per row = 630 FMA equivalents), + G once.  The 6 off-diagonal blocks: 7 iterations of 343 rows;
# the 4 diagonal blocks form one half (x-major) in 4 iterations of 322 + 3 x 301 rows (device/assemble.hpp, BLOCKS == 1)
SUMFACT_ASSEMBLY_FLOP_PER_ELEM = 2 * (6 * (7 * 343 * (112 + 441 + 630) + 343 * 16 * 7) + 4 * (4 * 343 * 112 + (322 + 3 * 301) * (441 + 630) + 343 * 16 * 7))


def algorithmic_bytes_per_dof(p, U, F=0):
    """read x 8 B + write y 8 B + connectivity 4*npe/(p^3 U) + vertices 192/(p^3 U) (+ 8F/U)."""
    npe = (p + 1) ** 3
    return 16.0 + (4.0 * npe + 192.0) / (p ** 3 * U) + 8.0 * F / U


KERNEL_SOURCES = ("l3ster_amd/csrc/device/sumfact_fast.hpp", "l3ster_amd/csrc/device/sumfact_apply.hpp",
                  "l3ster_amd/csrc/device/common.hpp", "l3ster_amd/csrc/user_kernels.hpp", "include/l3k/kernel_interface.hpp")
TRAFFIC_PROFILE = "profiles/r04_hbm_traffic.json"


def kernel_source_hash():
    """sha256 over the sources of the dominant kernel: a committed PMC measurement is only quoted for the kernel it was
    taken on."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()


def measured_traffic(ne, p):
    """(HBM bytes per launch, source) of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE /
    WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md prescribes).  The profile records the hash of the
    kernel sources it was taken at; for another workload or other sources the traffic is None (not measured)."""
    src = {"profile": TRAFFIC_PROFILE, "kernel_source_sha256": kernel_source_hash()}
    try:
        t = json.load(open(os.path.join(ROOT, TRAFFIC_PROFILE)))
        src["profile_kernel_source_sha256"] = t.get("kernel_source_sha256")
        if t["workload"] == f"{ne}x{ne}x{ne} order {p}" and t.get("kernel_source_sha256") == src["kernel_source_sha256"]:
            return t["traffic_bytes_per_launch"], dict(src, status="measured on these kernel sources")
        return None, dict(src, status="stale: profile taken on another workload or other kernel sources")
    except Exception as e:  # no profile committed
        return None, dict(src, status=f"no profile: {e.__class__.__name__}")


def cpu_baseline(part, mask, x_owned, p, U, applies=3):
    """The CPU oracle (port of the reference algorithm: per-element gather, sum-factorised sweeps in the reference's
    order, atomic scatter, element loop over the host threads) timed on the benchmark mesh itself with the benchmark's
    x (same mesh, same x, all host cores the affinity mask gives): `applies` applies, median."""
    import oracle_lib as O
    from helpers import oracle_mesh
    so = "/tmp/liboracle_native.so"
    try:
        O.build(so, march="native")
        L = O.lib(so)
        flavour = "-O3 -march=native"
    except Exception:  # pragma: no cover - fall back to the prebuilt x86-64-v3 library
        L = O.lib()
        flavour = "-O3 -march=x86-64-v3"
    # every core this job may use: the affinity mask, capped by the container's CPU quota where one is set (a GPU box
    # shows all of the host's CPUs in the mask but gives one GPU's job a share of them; threads beyond the share only
    # contend for the scatter's atomics)
    mask_cores = len(os.sched_getaffinity(0))
    cores = mask_cores
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(mask_cores, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    if os.environ.get("L3K_CPU_BASELINE_THREADS"):
        cores = int(os.environ["L3K_CPU_BASELINE_THREADS"])
    om = oracle_mesh(part, p + 1, U, np.arange(U), mask)
    x = np.asfortranarray(x_owned.reshape(-1, 1))
    y = np.zeros_like(x, order="F")
    times = []
    for _ in range(applies):
        t0 = time.perf_counter()
        O.mf_apply(om, 0, x, y, nthreads=cores, L=L)
        times.append(time.perf_counter() - t0)
    dofs = part.n_global_nodes * U
    dt = float(np.median(times))
    return {"value": dofs / dt, "unit": "DOF/s", "cores": cores, "kind": "port",
            "sample": f"median of {applies} applies on the benchmark mesh itself ({part.n_elems} order-{p} elements, {dofs} dofs, "
                      f"the benchmark's x), oracle/oracle.cpp orc_mf_apply {flavour}, {cores} threads "
                      f"(affinity mask: {mask_cores} CPUs, CPU quota applied where the container sets one), {' / '.join(f'{t:.2f}' for t in times)} s"}, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=15, help="untimed applies (the clock of an idle GPU ramps over the first ~10 launches)")
    ap.add_argument("--ne", type=int, default=64, help="elements per edge of each GPU's block")
    ap.add_argument("--order", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus not in PARTS:
        raise SystemExit("--gpus must be 1, 2, 4 or 8")
    if launch.needs_self_launch(args.gpus):  # typed without a launcher: this process starts the ranks and stays off the GPU
        return launch.self_launch(__file__, sys.argv[1:], args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path is the product; there is no CPU fallback)")
    # L3K_BENCH_REHEARSAL=1: all ranks on GPU 0 with the gloo backend -- the N > 1 code path (partition, exchange lists, split-phase
    # schedule, reductions over ranks, this JSON line) end to end on a one-GPU box; RCCL refuses two ranks on one device.  The
    # numbers of such a run mean nothing and the line says so ("rehearsal": true).
    rehearsal = os.environ.get("L3K_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
        if os.environ.get("L3K_BENCH_FAIL_RANK") == str(rank):  # (test hook of the self-launcher: a rank that dies before any collective)
            raise SystemExit(f"rank {rank}: failing on request (L3K_BENCH_FAIL_RANK)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("L3K_FORCE_DIST") == "1"  # the latter: smoke-test of the N > 1 code path at N = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    p, U, kid = args.order, 4, system.KERNEL_DIFFUSION3D
    parts = PARTS[world]
    ne_global = tuple(args.ne * q for q in parts)

    t_setup = time.perf_counter()
    part = system.CubePartition(ne_global, p, parts, rank, perturb=0.1)
    mask = part.dirichlet_mask(U)
    ctx = system.Context(local_rank, torch.cuda.current_stream().cuda_stream)
    ctx.set_tuning(generic_below=0)  # small --ne runs time the kernel named in the roofline object, not the small-launch route
    mesh = system.DeviceMesh(ctx, part, U, mask)
    mf = system.MatrixFreeSystem(mesh, kid, [1.0, 1.0])
    n_owned = part.n_owned_nodes * U
    X = system.synthetic_vector_torch(part.node_grid_id[:part.n_owned_nodes], U, dev)
    Y = torch.empty_like(X)
    op, native = None, False
    if use_dist:
        # Transport of the ghost exchange.  Default: torch.distributed P2P (batch_isend_irecv = RCCL send / receive over xGMI)
        # around the split-phase C-ABI calls.  L3K_BENCH_TRANSPORT=native: the exchange inside the library (l3k_mf_apply_dist:
        # RCCL groups on the library's own stream).  Its schedule is tested with several ranks through the in-process
        # transport (tests/test_gpu_dist_cabi.py), but RCCL under it has only ever run as a self exchange on one GPU, so it
        # stays opt-in until a run on two or more GPUs is on record.  Set-up in two steps so that no rank can fall out of a
        # collective: (1) every rank checks locally, without communication, that RCCL loads and that its exchange lists are
        # what the library takes; all ranks agree (MIN); (2) only then the collective part (broadcast of the unique id,
        # ncclCommInitRank) runs, on all ranks or on none.
        if os.environ.get("L3K_BENCH_TRANSPORT", "torch") == "native":
            ok = 1
            try:
                NativeHalo.check_local(part)
            except Exception as exc:  # pragma: no cover
                ok = 0
                print(f"[bench rank {rank}] native halo unavailable ({exc}); torch.distributed transport", file=sys.stderr, flush=True)
            agree = torch.tensor([ok], dtype=torch.int32, device="cpu" if rehearsal else dev)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            native = bool(agree.item())
        if native:
            op = NativeDistributedOperator(mf, NativeHalo(ctx, part, U, rank, world))
        else:
            op = DistributedOperator(mf, HaloPlan(part, U, dev), transport=HostStagedTransport() if rehearsal else None)
    t_setup = time.perf_counter() - t_setup

    n_launches = 3 if op is not None else 1  # partitioned: first interior half, border elements, second interior half
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_launches)] for _ in range(args.steps)]

    def step(i=None):
        if op is not None and native:
            op.apply(X, Y, 1.0, 0.0)  # (timed steps: the library stamps its three element launches, timing_begin below)
        elif op is not None:
            op.apply(X, Y, 1.0, 0.0, events=None if i is None else ev[i])
        elif i is None:
            mf.apply(X, Y, 1.0, 0.0)
        else:  # same three launches as l3k_mf_apply, with events around the element kernel
            mf.scale(Y, 0.0)
            ev[i][0][0].record()
            mf.apply_elems(2, X, None, Y, None, 1.0, 0.0)
            ev[i][0][1].record()
            mf.dirichlet_rows(X, Y, 1.0)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if native:
        op.timing_begin(args.steps)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # the dominant kernel, HIP events on the launch stream inside the timed region: one launch over all elements (one GPU), or
    # the three launches of a partitioned apply (first interior half, border elements, second interior half) summed -- per
    # timed step the MAX over the ranks (the ranks that own Dirichlet faces run longer)
    if native:
        kernel_times = [sum(op.timing_get(i)) for i in range(args.steps)]
    else:
        kernel_times = [sum(a.elapsed_time(b) for a, b in e) for e in ev]
    if use_dist:
        kt = torch.tensor(kernel_times, dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(kt, op=dist.ReduceOp.MAX)
        kernel_times = kt.tolist()
    ms = float(np.median(kernel_times))  # median of the timed steps
    global_dofs = part.n_global_nodes * U
    value = global_dofs * args.steps / elapsed
    bpd = algorithmic_bytes_per_dof(p, U)
    result = {
        "metric": "DOF/s for MF operator apply (Diffusion3D, hex p=6)" if p == 6 else f"DOF/s for MF operator apply (Diffusion3D, hex p={p})",
        "value": value, "unit": "DOF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", **({"rehearsal": True} if rehearsal else {}),
        "config": {"workload": f"Diffusion3D matrix-free sum-factorised apply, hex mesh {ne_global[0]}x{ne_global[1]}x{ne_global[2]}, "
                               f"order {p}, U=4 E=7, {global_dofs} global dofs, {args.ne}^3 elements per GPU, "
                               f"partition {parts[0]}x{parts[1]}x{parts[2]}",
                   "alpha": 1.0, "beta": 0.0, "setup_s": round(t_setup, 2)},
    }
    if rank == 0:
        # algorithmic bytes = 17.81 B per dof of the elements one rank's launches process
        n_launch_elems = part.n_elems
        launch_dofs = n_launch_elems * p ** 3 * U if op is not None else global_dofs
        alg_bytes = bpd * launch_dofs
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        flop_per_elem = FP64_FLOP_PER_ELEM.get(p)
        traffic, traffic_source = measured_traffic(args.ne, p) if world == 1 else (None, None)
        result["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBS,
                              "traffic": traffic, "traffic_source": traffic_source,
                              # the kernel the timed launches took, as the launcher itself reports it (l3k_mf_route)
                              "kernel": mf.route(2 if op is None else 0) + (" (the three element launches of the partitioned apply, summed; per step the max over the ranks)" if op is not None else ""),
                              "transport": None if op is None else ("l3k_mf_apply_dist (RCCL inside the library)" if native else "torch.distributed P2P"),
                              "kernel_ms": ms, "kernel_ms_mean": float(np.mean(kernel_times)), "kernel_ms_stat": "median of the timed launches",
                              "bytes_per_dof": bpd, "dofs_per_launch": launch_dofs,
                              "algorithmic_bytes_per_launch": alg_bytes,
                              "fp64_note": "not HBM-bound (DESIGN.md 4.1): the kernel is bound by FP64 vector issue at the occupancy its "
                                           "registers and LDS admit; fp64_frac = executed vector FP64 flops per element (ISA count) against "
                                           "the 78.6 TFLOP/s FP64 peak (vector and matrix FP64 share one pipe on this part).  Perfect issue "
                                           "of this instruction stream on 49 of 64 lanes would be ~0.34 of the HBM roofline; the practical "
                                           "ceiling of one wave per element at 7-8 waves per CU is what is measured here",
                              "fp64_tflops": None if flop_per_elem is None else flop_per_elem * n_launch_elems / (ms * 1e-3) / 1e12,
                              "fp64_peak_tflops": 78.6,
                              "fp64_frac": None if flop_per_elem is None else flop_per_elem * n_launch_elems / (ms * 1e-3) / 1e12 / 78.6}
        if world == 1 and op is None and p == 6:
            # the second half of BASELINE.json's metric: element matrices/s of LocalAssembly, order 6, streaming mode
            # (checksums instead of 15 MB per matrix); outside the timed region.  Default algorithm: sum-factorised assembly
            # (device/assemble.hpp: ~47 MFLOP per element on the vector pipe); beside it the dense K_e = (W Z)^T Z product on
            # the FP64 matrix cores (4 523 MFLOP per element using symmetry), the formulation the reference computes
            # sum-factorised kernel: the FULL sweep over the benchmark mesh (args.ne^3 elements; 262 144 at 64^3), in batches
            # of 2048; the dense product (28x slower) on three batches of 512 of the same mesh
            batch = 2048  # (298 k matrices/s against 286 k at 512: fewer launch tails; workspace 635 MB)
            amf = mf

            def assembly_rate(n_elems, batch=batch):
                amf.local_assemble(0, batch, want_K=False, want_F=False, want_checksum=True)
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record()
                for first in range(0, n_elems, batch):
                    amf.local_assemble(first, min(batch, n_elems - first), want_K=False, want_F=False, want_checksum=True)
                a1.record()
                torch.cuda.synchronize()
                return n_elems / (a0.elapsed_time(a1) * 1e-3)

            n_sweep = part.n_elems
            rate = assembly_rate(n_sweep)
            with ctx.tuning(assemble_dense=1):
                rate_dense = assembly_rate(min(1536, n_sweep), 512)
            # the reference's return value itself: row-major K_e stored in HBM, symmetric bit for bit (AssembleLocalSystem.hpp:168-182), 256
            # matrices (x-major tiled assembly of the lower triangle + the mirroring transposition: DESIGN.md 4.4)
            n_st = min(256, n_sweep)
            Kst = torch.empty((n_st, (p + 1) ** 3 * U, (p + 1) ** 3 * U), dtype=torch.float64, device=dev)
            amf.local_assemble_into(Kst, 0, n_st)
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            for _ in range(3):
                amf.local_assemble_into(Kst, 0, n_st)
            s1.record()
            torch.cuda.synchronize()
            rate_stored = 3 * n_st / (s0.elapsed_time(s1) * 1e-3)
            stored_symmetric = bool(torch.equal(Kst[:4], Kst[:4].transpose(1, 2)))
            del Kst
            nd, kd = (p + 1) ** 3 * U, (p + 1) ** 3 * 7
            dense_flops = kd * nd * (nd + 1)          # symmetric half of 2*K*N^2
            sf_flops = SUMFACT_ASSEMBLY_FLOP_PER_ELEM  # executed by the sum-factorised kernel (order 6, U = 4, E = 7)
            result["assembled_path"] = {"metric": "element-matrices/s for assembled path (LocalAssembly, Diffusion3D, hex p=6)",
                                        "value": rate, "unit": "element matrices/s", "batch": batch,
                                        "sample": f"full streaming sweep over the benchmark mesh: {n_sweep} elements (checksums on the device)",
                                        "algorithm": "sum-factorised assembly on index pairs (O(n^7) per pair of unknowns), FP64 vector pipe",
                                        "roofline": {"bound": "fp64-valu", "achieved": rate * sf_flops / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                                                     "frac": rate * sf_flops / 1e12 / 78.6,
                                                     "flops": "executed: %.1f MFLOP per element; FP64 matrix and vector pipes are one pipe "
                                                              "on this part (78.6 TFLOP/s, profiles/r02_fp64_vector_matrix_coexecution.log)"
                                                              % (sf_flops / 1e6),
                                                     "dense_equivalent_tflops": rate * dense_flops / 1e12},
                                        "stored_row_major": {"value": rate_stored, "unit": "element matrices/s", "batch": n_st,
                                                             "bitwise_symmetric": stored_symmetric,
                                                             "GB_per_s_of_matrices": rate_stored * nd * nd * 8 / 1e9},
                                        "dense_mfma_kernel": {"value": rate_dense, "unit": "element matrices/s",
                                                              "roofline": {"bound": "mfma", "achieved": rate_dense * dense_flops / 1e12,
                                                                           "peak": 78.6, "unit": "TFLOP/s",
                                                                           "frac": rate_dense * dense_flops / 1e12 / 78.6,
                                                                           "flops": "symmetric half, 2*K*N*(N+1)/2 per element"}}}
            del amf
            # BASELINE.json configs[1]: the same apply at order 4 on the same 64^3 mesh (outside the timed region)
            p4 = 4
            part4 = system.CubePartition(args.ne, p4, perturb=0.1)
            mf4 = system.MatrixFreeSystem(system.DeviceMesh(ctx, part4, U, part4.dirichlet_mask(U)), kid, [1.0, 1.0])
            X4 = system.synthetic_vector_torch(part4.node_grid_id[:part4.n_owned_nodes], U, dev)
            Y4 = torch.empty_like(X4)
            for _ in range(args.warmup):
                mf4.apply(X4, Y4, 1.0, 0.0)
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0.record()
            for _ in range(args.steps):
                mf4.apply(X4, Y4, 1.0, 0.0)
            b1.record()
            torch.cuda.synchronize()
            ms4 = b0.elapsed_time(b1) / args.steps
            dofs4 = part4.n_global_nodes * U
            result["order4_apply"] = {"workload": f"Diffusion3D matrix-free apply, hex mesh {args.ne}^3, order 4, {dofs4} dofs "
                                                  "(BASELINE.json configs[1])",
                                      "value": dofs4 / (ms4 * 1e-3), "unit": "DOF/s", "ms_per_step": ms4, "kernel": mf4.route(),
                                      "roofline_frac_hbm": dofs4 / (ms4 * 1e-3) * algorithmic_bytes_per_dof(p4, U) / 1e9 / HBM_PEAK_GBS}
            del mf4, X4, Y4, part4
            # the reference's own benchmark mesh is the UNIFORM cube (benchmarks/Diffusion3D: makeCubeMesh): every element is a
            # parallelepiped, which the library detects at mesh creation and serves with the kernel variant that inverts one
            # Jacobian per element instead of one per quadrature point.  The headline above stays on the perturbed mesh
            # (general tri-linear geometry); this is the same apply on the uniform one, outside the timed region
            partu = system.CubePartition(args.ne, p, perturb=0.0)
            mfu = system.MatrixFreeSystem(system.DeviceMesh(ctx, partu, U, partu.dirichlet_mask(U)), kid, [1.0, 1.0])
            Yu = torch.empty_like(X)
            for _ in range(args.warmup):
                mfu.apply(X, Yu, 1.0, 0.0)
            u0, u1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            u0.record()
            for _ in range(args.steps):
                mfu.apply(X, Yu, 1.0, 0.0)
            u1.record()
            torch.cuda.synchronize()
            msu = u0.elapsed_time(u1) / args.steps
            result["uniform_mesh_apply"] = {"workload": f"the same apply on the unperturbed {args.ne}^3 cube (all elements affine: the reference benchmark's mesh)",
                                            "value": global_dofs / (msu * 1e-3), "unit": "DOF/s", "ms_per_step": msu, "kernel": mfu.route(),
                                            "roofline_frac_hbm_whole_apply": global_dofs / (msu * 1e-3) * bpd / 1e9 / HBM_PEAK_GBS}
            del mfu, Yu, partu
        if world == 1 and op is None:
            if not args.no_cpu_baseline:
                # the CPU run doubles as a full-size parity check: the whole output vector of the timed GPU launches
                # (dynamic XCD-chunked batch distribution included) against the oracle on the same mesh and x
                base, y_cpu = cpu_baseline(part, mask, X.cpu().numpy().reshape(-1), p, U)
                result["cpu_baseline"] = base
                y_gpu = Y.cpu().numpy().reshape(-1)
                err = float(np.linalg.norm(y_gpu - y_cpu[:, 0]) / np.linalg.norm(y_cpu[:, 0]))
                result["config"]["parity_vs_oracle_rel_l2"] = err
                result["config"]["parity_sample"] = "the whole output vector of the benchmark mesh"
                if not err < 1e-11:
                    raise SystemExit(f"GPU result differs from the oracle: rel L2 {err}")
    if rehearsal and use_dist:
        # the partitioned result against ONE rank's apply on the whole mesh (same global numbering, perturbation and x): x^T A x
        # over the owned rows of all ranks, and the norm of y
        loc = torch.stack([(X * Y).sum(), (Y * Y).sum()]).cpu()
        dist.all_reduce(loc)
        if rank == 0:
            whole = system.CubePartition(ne_global, p, perturb=0.1)
            mfw = system.MatrixFreeSystem(system.DeviceMesh(ctx, whole, U, whole.dirichlet_mask(U)), kid, [1.0, 1.0])
            Xw = system.synthetic_vector_torch(whole.node_grid_id[:whole.n_owned_nodes], U, dev)
            Yw = torch.empty_like(Xw)
            mfw.apply(Xw, Yw, 1.0, 0.0)
            ref = torch.stack([(Xw * Yw).sum(), (Yw * Yw).sum()]).cpu()
            rel = ((loc - ref).abs() / ref.abs()).max().item()
            result["rehearsal_check"] = {"xAx_and_yy_vs_one_rank_rel": rel, "xAx": loc[0].item()}
            if not rel < 1e-11:
                raise SystemExit(f"partitioned apply differs from the one-rank apply on the whole mesh: {rel}")
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
