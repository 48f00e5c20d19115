"""ad-hoc timing sweep (developer tool, not the contract bench)"""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cfs_spmv_amd as cfs
from cfs_spmv_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "Flan_1565"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
configs = sys.argv[3:] or ["2560,256"]
t = time.time()
n, rp, ci, va, low = synth.generate(name, scale)
print(f"gen {name} n={n} nnz_full={rp[-1]} nnz_low={low} {time.time()-t:.1f}s", flush=True)
import os
f32 = os.environ.get("QB_DTYPE", "f64") == "f32"
if f32:
    va = va.astype(np.float32)
x = torch.from_numpy(synth.make_x(n, 42, np.float32 if f32 else np.float64)).cuda()
y = torch.empty(n, dtype=x.dtype, device="cuda")
if os.environ.get("QB_CSR"):
    G = cfs.CsrMatrix(n, n, rp, ci, va)
    for _ in range(3):
        G.dense_vector_multiply(y, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        G.dense_vector_multiply(y, x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    b = int(rp[-1]) * (4 + va.itemsize) + n * (4 + 2 * va.itemsize)
    print(json.dumps(dict(cfg="csr", ms=round(ms, 4), GBs=round(b / ms / 1e6, 1), frac=round(b / ms / 1e6 / 8000, 3))), flush=True)
    G.close()
for cfg in configs:
    slots, block = (int(v) for v in cfg.split(",")[:2])
    mtn = int(cfg.split(",")[2]) if cfg.count(",") >= 2 else 0
    flags = int(cfg.split(",")[3]) if cfg.count(",") >= 3 else 0
    t = time.time()
    A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(slots, mtn, block, flags))
    st = A.stats()
    tp = time.time() - t
    for _ in range(5):
        A.dense_vector_multiply(y, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 50
    e0.record()
    for _ in range(iters):
        A.dense_vector_multiply(y, x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    e0.record()
    for _ in range(iters):
        A.spmv_phases(y, x, None, 1)
    e1.record(); torch.cuda.synchronize()
    ms_tile = e0.elapsed_time(e1) / iters
    alg = st["bytes_algorithmic"]
    print(json.dumps(dict(cfg=cfg, ms=round(ms, 4), tile_ms=round(ms_tile, 4), tile_frac=round(alg / ms_tile / 1e6 / 8000, 3), alg_GBs=round(alg / ms / 1e6, 1),
                          frac=round(alg / ms / 1e6 / 8000, 3), streamed_GBs=round(st["bytes_streamed"] / ms / 1e6, 1),
                          gflops=round(2 * st["nnz_full"] / ms / 1e6, 1), tiles=st["ntiles"], halo=st["halo_slots"],
                          lds=st["lds_bytes"], preproc_s=round(tp, 2))), flush=True)
    A.close()
