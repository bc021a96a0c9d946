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
