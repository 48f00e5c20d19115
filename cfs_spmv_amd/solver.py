"""Solver-style caller of the SpMV path: conjugate gradients on resident vectors.

SURVEY.md 8(f)-4 / 8(e): the reference's only caller is a benchmark loop with a
fixed x; a solver feeds every product back as the next input, which is what the
y -> x all-gather of the sharded path exists for.  This is host-side control
flow only: the matrix-vector product is the HIP path (SymMatrix / ShardedSym),
the vector updates and dot products are torch plumbing on the same device.
"""
import math


def cg(A, b, tol=1e-10, maxiter=1000, x0=None):
    """solve A u = b for a symmetric positive definite SymMatrix (one GPU).
    b, u: device tensors of A.nrows() values.  Returns (u, iterations, relative
    residual ||b - A u|| / ||b||), the residual recomputed at the end."""
    import torch
    u = torch.zeros_like(b) if x0 is None else x0.clone()
    q = torch.empty_like(b)
    A.dense_vector_multiply(q, u)
    r = b - q
    p = r.clone()
    rr = float(torch.dot(r, r))
    bnorm = math.sqrt(float(torch.dot(b, b))) or 1.0
    it = 0
    while it < maxiter and math.sqrt(rr) > tol * bnorm:
        A.dense_vector_multiply(q, p)          # the hot path
        alpha = rr / float(torch.dot(p, q))
        u.add_(p, alpha=alpha)
        r.add_(q, alpha=-alpha)
        rr_new = float(torch.dot(r, r))
        p.mul_(rr_new / rr).add_(r)
        rr = rr_new
        it += 1
    A.dense_vector_multiply(q, u)
    res = math.sqrt(float(torch.dot(b - q, b - q))) / bnorm
    return u, it, res


def cg_native(A, b, tol=1e-10, maxiter=1000, x0=None, check_every=8):
    """the same iteration inside the library (cfs_hip_sym_cg, cfs_spmv_amd/csrc/cfs_solver.hpp):
    five launches per iteration and no host round trip -- the loop above reads two dot products
    on the host per iteration, a stream synchronisation each.  Returns (u, iterations, relative
    residual), like cg()."""
    import torch
    u = torch.zeros_like(b) if x0 is None else x0.clone()
    it, res = A.cg(u, b, tol=tol, maxiter=maxiter, check_every=check_every)
    return u, it, res


def cg_sharded(S, row_splits, b_block, tol=1e-10, maxiter=1000):
    """the same iteration over 1-D row blocks (cfs_spmv_amd.dist.ShardedSym): every
    rank keeps its block of u, r, q and a full replica of the search direction p,
    refreshed by one all-gather per iteration; dot products are all-reduced.
    Returns (u_block, iterations, relative residual)."""
    torch, dist = S.torch, S.dist
    S.setup_allgather(row_splits)
    rb, re = int(row_splits[S.rank]), int(row_splits[S.rank + 1])
    n = int(row_splits[-1])
    dev = b_block.device

    def gdot(a, c):
        t = torch.dot(a, c).reshape(1)
        if S.stage or t.device.type == "cpu":
            t = t.cpu()
            dist.all_reduce(t, group=S.pg)
        else:
            dist.all_reduce(t, group=S.pg)
        return float(t)

    u = torch.zeros_like(b_block)
    r = b_block.clone()                         # u = 0: r = b
    p_full = torch.zeros(n, dtype=b_block.dtype, device=dev)
    S.allgather_rows(r, p_full)
    q = torch.empty_like(b_block)
    rr = gdot(r, r)
    bnorm = math.sqrt(rr) or 1.0
    it = 0
    while it < maxiter and math.sqrt(rr) > tol * bnorm:
        S.spmv(q, p_full)                       # the hot path, exchange included
        p = p_full[rb:re]
        alpha = rr / gdot(p, q)
        u.add_(p, alpha=alpha)
        r.add_(q, alpha=-alpha)
        rr_new = gdot(r, r)
        p_new = r + p * (rr_new / rr)
        S.allgather_rows(p_new, p_full)         # y -> x: next input on every rank
        rr = rr_new
        it += 1
    u_full = torch.zeros(n, dtype=b_block.dtype, device=dev)
    S.allgather_rows(u, u_full)
    S.spmv(q, u_full)
    d = b_block - q
    res = math.sqrt(gdot(d, d)) / bnorm
    return u, it, res
