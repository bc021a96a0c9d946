"""Jacobi-preconditioned conjugate gradients around the matrix-free operator.

The reference delegates the iteration to Trilinos Belos ("Block CG", solve/BelosSolvers.hpp:116-122) with its own native
Jacobi preconditioner (solve/NativePreconditioners.hpp:36-96); Belos is a third-party dependency that is not under
/root/reference, so the CG arithmetic is restated from the published algorithm (Hestenes-Stiefel PCG) and parity is
pinned end to end (solution / error thresholds), not iterate by iterate.  Vector updates and dot
products are torch ops (plumbing); the operator apply -- the hot path -- is the HIP kernel behind `apply`.
"""
import torch
import torch.distributed as dist


def jacobi_inverse(diag, damping=1.0, threshold=0.0):
    """NativeJacobiImpl::init (solve/NativePreconditioners.hpp:75-96): sign(d)*damping / max(|d|, threshold)."""
    sign = torch.where(diag < 0, -torch.ones_like(diag), torch.ones_like(diag))
    return sign * damping / torch.clamp(diag.abs(), min=threshold)


class IterSolveResult:
    def __init__(self, tol, num_iters, converged):
        self.tol, self.num_iters, self.converged = tol, num_iters, converged

    def __repr__(self):
        return f"IterSolveResult(tol={self.tol:.3e}, num_iters={self.num_iters}, converged={self.converged})"


def cg(apply, b, x, minv=None, tol=1e-6, max_iters=10_000, residual_scaling="none", group=None, throw_on_fail=True):
    """Solves A x = b for one column (1-D tensors over the OWNED rows of this rank); x holds the initial guess and the
    result.  apply(p, out) computes out <- A p.  Options mirror IterSolverOpts (solve/SolverInterface.hpp:26-37):
    residual_scaling in {"none", "initial", "rhs"}.  `group`: torch.distributed group for the dot products of a
    partitioned vector (None = single rank)."""

    def dot(u, v):
        s = torch.dot(u, v)
        if group is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
        return s.item()

    r = torch.empty_like(b)
    apply(x, r)
    r = b - r
    rr0 = dot(r, r) ** 0.5
    scale = {"none": 1.0, "initial": rr0 if rr0 > 0 else 1.0, "rhs": max(dot(b, b) ** 0.5, 1e-300)}[residual_scaling]
    z = r * minv if minv is not None else r.clone()
    p = z.clone()
    rz = dot(r, z)
    ap = torch.empty_like(b)
    res = rr0 / scale
    it = 0
    while res > tol and it < max_iters:
        apply(p, ap)
        alpha = rz / dot(p, ap)
        x.add_(p, alpha=alpha)
        r.sub_(ap, alpha=alpha)
        res = dot(r, r) ** 0.5 / scale
        it += 1
        if res <= tol:
            break
        z = r * minv if minv is not None else r
        rz_new = dot(r, z)
        p.mul_(rz_new / rz).add_(z)
        rz = rz_new
    converged = res <= tol
    if throw_on_fail and not converged:
        raise RuntimeError("Solver failed to converge")  # solve/BelosSolvers.hpp:103
    return IterSolveResult(res, it, converged)


def jacobi_inverse_native(ctx, diag, damping=1.0, threshold=0.0):
    """NativeJacobiImpl::init through the C ABI (l3k_jacobi_inverse)."""
    import ctypes as C
    from . import capi
    out = torch.empty_like(diag)
    capi.check(capi.load().l3k_jacobi_inverse(ctx._h, C.c_void_p(diag.data_ptr()), diag.numel(), float(damping),
                                              float(threshold), C.c_void_p(out.data_ptr())))
    return out


_SCALING = {"none": 0, "initial": 1, "rhs": 2}


def pcg(system, b, x, minv=None, tol=1e-6, max_iters=10_000, residual_scaling="none", check_every=1, throw_on_fail=True):
    """Jacobi-PCG entirely behind the C ABI (l3k_pcg_solve): apply, fused vector updates and reductions run on the
    context's stream, the host only reads 32 bytes per convergence check.  Single rank; `system` is a
    l3ster_amd.system.MatrixFreeSystem, b / x / minv 1-D device tensors over its owned dofs."""
    import ctypes as C
    from . import capi
    opts = capi.CgOpts(float(tol), int(max_iters), _SCALING[residual_scaling], int(check_every))
    if b.dim() == 2:  # a multivector (ncols, ld) of right-hand sides: the columns one after the other (l3k_pcg_solve_cols)
        nc = b.shape[0]
        if x.shape != b.shape or b.stride(1) != 1 or x.stride(1) != 1:
            raise capi.L3KError("b and x must be (ncols, ld) tensors of one shape with unit stride along rows")
        res_c = (capi.CgResult * nc)()
        capi.check(capi.load().l3k_pcg_solve_cols(system._h, C.c_void_p(b.data_ptr()), b.stride(0) if nc > 1 else b.shape[1],
                                                  C.c_void_p(x.data_ptr()), x.stride(0) if nc > 1 else x.shape[1], nc,
                                                  C.c_void_p(0 if minv is None else minv.data_ptr()), C.byref(opts), res_c))
        out = [IterSolveResult(r.achieved_tol, r.iterations, bool(r.converged)) for r in res_c]
        if throw_on_fail and not all(r.converged for r in out):
            raise RuntimeError("Solver failed to converge")  # solve/BelosSolvers.hpp:103
        return out
    res = capi.CgResult()
    capi.check(capi.load().l3k_pcg_solve(system._h, C.c_void_p(b.data_ptr()), C.c_void_p(x.data_ptr()),
                                         C.c_void_p(0 if minv is None else minv.data_ptr()), C.byref(opts), C.byref(res)))
    if throw_on_fail and not res.converged:
        raise RuntimeError("Solver failed to converge")  # solve/BelosSolvers.hpp:103
    return IterSolveResult(res.achieved_tol, res.iterations, bool(res.converged))


def pcg_distributed(op, ctx, b, x, minv=None, tol=1e-6, max_iters=10_000, residual_scaling="none", group=None,
                    throw_on_fail=True, allreduce=None, check_every=1):
    """The same iteration for a partitioned system: `op.apply(X, Y)` is a DistributedOperator over this rank's owned
    rows; the fused l3k_cg_* kernels keep the scalars in a device block that is all-reduced between them (two small
    all-reduces per iteration, as Belos does)."""
    import ctypes as C
    from . import capi
    lib = capi.load()
    n = b.numel()
    vp = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    s = torch.zeros(8, dtype=torch.float64, device=b.device)
    r, p, ap = torch.empty_like(b), torch.empty_like(b), torch.empty_like(b)
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    import inspect
    fuse_energy = "energy" in inspect.signature(op.apply).parameters

    def reduce(view):
        if allreduce is not None:  # pluggable (the threaded multi-rank emulation of the tests)
            allreduce(view)
        elif multi:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group)

    op.apply(x[None, :], r[None, :])
    capi.check(lib.l3k_cg_init(ctx._h, vp(r), vp(b), vp(p), vp(minv), n, vp(s)))
    reduce(s[2:4])
    s[0] = s[2]
    bb = torch.dot(b, b).reshape(1)
    reduce(bb)
    h = s[:4].tolist()
    rr0 = h[3] ** 0.5
    scale = {"none": 1.0, "initial": rr0 if rr0 > 0 else 1.0, "rhs": max(bb.item() ** 0.5, 1e-300)}[residual_scaling]
    res, it = rr0 / scale, 0
    while res > tol and it < max_iters:
        if fuse_energy:  # <p, A p> from the element kernels' quadrature stage where they can (l3k_mf_energy_*)
            op.apply(p[None, :], ap[None, :], energy=s)
            if it == 0:
                # element-wise shares (fused) and row-wise shares (dot product) of <p, A p> do not add up across ranks:
                # every rank must take the same route.  Decided once -- it depends on the launch sizes only.
                n_not = torch.tensor([0.0 if op.energy_fused else 1.0], dtype=torch.float64, device=b.device)
                reduce(n_not)
                fuse_energy = n_not.item() == 0.0
            elif not op.energy_fused:
                raise RuntimeError("the element kernels stopped accumulating <p, A p>")
        else:
            op.apply(p[None, :], ap[None, :])
        if not fuse_energy:
            capi.check(lib.l3k_cg_dot_pap(ctx._h, vp(p), vp(ap), n, vp(s)))
        reduce(s[1:2])
        capi.check(lib.l3k_cg_update_z(ctx._h, vp(r), vp(ap), vp(minv), n, vp(s)))  # (r holds z = M^-1 r: l3k.h)
        reduce(s[2:4])
        capi.check(lib.l3k_cg_update_px(ctx._h, vp(p), vp(x), vp(r), n, vp(s)))
        it += 1
        if it % check_every == 0 or it == max_iters:  # (the only host synchronisation of the iteration)
            res = s[3].item() ** 0.5 / scale
    converged = res <= tol
    if throw_on_fail and not converged:
        raise RuntimeError("Solver failed to converge")
    return IterSolveResult(res, it, converged)
