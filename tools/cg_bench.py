"""per-iteration time of conjugate gradients on resident vectors: the torch-driven loop of
cfs_spmv_amd/solver.py (two host-read dot products per iteration) against cfs_hip_sym_cg (the whole
iteration behind the C ABI, no host round trip).  Fixed number of iterations (tol = 0).
usage: python tools/cg_bench.py [matrix[:scale] ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cfs_spmv_amd as cfs
from cfs_spmv_amd import synth
from cfs_spmv_amd.solver import cg, cg_native

out = {}
for spec in (sys.argv[1:] or ["pwtk", "ldoor", "Flan_1565"]):
    name, _, sc = spec.partition(":")
    n, rp, ci, va, _ = synth.generate(name, float(sc or 1.0))
    A = cfs.SymMatrix(n, rp, ci, va)
    b = torch.from_numpy(synth.make_x(n, 11)).cuda()
    y = torch.empty_like(b)
    for _ in range(50):
        A.dense_vector_multiply(y, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        A.dense_vector_multiply(y, b)
    torch.cuda.synchronize()
    spmv_us = (time.perf_counter() - t0) / 200 * 1e6
    K = 200
    res = {"n": n, "spmv_us": round(spmv_us, 2)}
    side = torch.cuda.Stream()  # (a stream capture needs a stream of its own: CFS_HIP_CG_GRAPH=1)

    def on_side(f):
        def run():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                out = f()
            torch.cuda.current_stream().wait_stream(side)
            return out
        return run
    for label, fn in (("torch_loop", lambda: cg(A, b, tol=0.0, maxiter=K)),
                      ("native_side_stream", on_side(lambda: cg_native(A, b, tol=0.0, maxiter=K, check_every=16))),
                      ("native_check8", lambda: cg_native(A, b, tol=0.0, maxiter=K, check_every=8)),
                      ("native_check16", lambda: cg_native(A, b, tol=0.0, maxiter=K, check_every=16))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        u, it, r = fn()
        torch.cuda.synchronize()
        res[label + "_us_per_iteration"] = round((time.perf_counter() - t0) / max(it, 1) * 1e6, 2)
        res[label + "_iterations"] = it
    out[spec] = res
    A.close()
print(json.dumps(out))
