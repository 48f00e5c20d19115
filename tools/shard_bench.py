"""developer tool: time ONE rank's mirrored shard of an N-way split on this GPU
(predicts the per-rank step of the driver's multi-GPU run; no process group)
usage: shard_bench.py <matrix> <scale> <N> [ranks...]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cfs_spmv_amd as cfs
from cfs_spmv_amd import synth
name, scale, N = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
ranks = [int(a) for a in sys.argv[4:]] or list(range(N))
flags = int(os.environ.get("SB_FLAGS", "0"))
slots, block = int(os.environ.get("SB_SLOTS", "0")), int(os.environ.get("SB_BLOCK", "0"))
n, rp, ci, va, low = synth.generate(name, scale)
if os.environ.get("QB_DTYPE") == "f32":
    va = va.astype(np.float32)
x = torch.from_numpy(synth.make_x(n, 42, va.dtype)).cuda()
rs = cfs.balanced_splits(n, rp, ci, N)
for r in ranks:
    t0 = time.time()
    A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(slots, 0, block, flags), row_splits=rs, rank=r)
    pre = time.time() - t0
    st = A.stats()
    y = torch.empty(st["row_end"] - st["row_begin"], dtype=x.dtype, device="cuda")
    for _ in range(10):
        A.spmv_phases(y, x, None, 3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = {}
    for ph, key in ((3, "ms"), (1, "tile_ms")):
        e0.record()
        for _ in range(100):
            A.spmv_phases(y, x, None, ph)
        e1.record(); torch.cuda.synchronize()
        res[key] = round(e0.elapsed_time(e1) / 100, 4)
    alg = st["bytes_algorithmic"]
    print(json.dumps(dict(N=N, rank=r, rows=st["row_end"] - st["row_begin"], nnz_low=st["nnz_low"],
                          mirror=st["mirror_entries"], tiles=st["ntiles"], halo=st["halo_slots"],
                          fold_rows=st["fold_rows"], lds=st["lds_bytes"], **res,
                          tile_frac=round(alg / res["tile_ms"] / 1e6 / 8000, 3),
                          whole_matrix_gflops_if_all_ranks_like_this=round(2.0 * int(rp[-1]) / res["ms"] / 1e6, 1),
                          preproc_s=round(pre, 2))), flush=True)
    A.close()
