"""device-memory leak check: create / multiply / destroy every kind of handle a few times and compare the
memory in use (run on the GPU box: python tools/leak_check.py).  Round 3: 842 MiB after the warm-up cycles,
842 MiB after six more."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cfs_spmv_amd as cfs
from cfs_spmv_amd import synth
torch.cuda.init()
n, rp, ci, va, _ = synth.generate("ldoor", 0.1)
x = torch.from_numpy(synth.make_x(n, 1)).cuda(); y = torch.empty_like(x)
def used():
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return (t - f) / 2**20
def cycle(k):
    for flags in (0, 128, 1024, 128 | 1024, 2048):
        A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=flags | (32 if k % 2 else 0)))
        A.dense_vector_multiply(y, x); A.close()
    G = cfs.CsrMatrix(n, n, rp, ci, va); G.dense_vector_multiply(y, x); G.close()
    A = cfs.SymMatrix(n, rp, ci, va); u = torch.zeros_like(x); A.cg(u, x, maxiter=5); A.close()
    M = cfs.SymMatrix(n, rp, ci, va, ngpus=2); M.dense_vector_multiply(y, x); M.close()
cycle(0); cycle(1)
base = used()
for k in range(6): cycle(k)
print("MiB in use: after warm-up %.0f, after 6 more cycles %.0f" % (base, used()))
