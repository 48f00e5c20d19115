"""One process, N row blocks (cfs_hip_sym_create_multi_*: what CSRMatrix::tune builds for
CFS_NUM_GPUS=N): ms per SpMV of the native forms of the one-process multi-device handle --
mirrored shards with x over peer access / replicated x, and the exchange form with one native
reduce-scatter per SpMV (cfs_hip_comm_*).  On a one-GPU box the N shards share the device (a
REHEARSAL of the code paths: copies are device-local, the collective runs on the peer
transport); on an N-GPU node the same script times the real thing.
usage: multi_bench.py [matrix] [scale] [ngpus ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cfs_spmv_amd as cfs
from cfs_spmv_amd import _lib, synth

name = sys.argv[1] if len(sys.argv) > 1 else "Flan_1565"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
counts = [int(v) for v in sys.argv[3:]] or [2, 4, 8]
torch.cuda.init()
ndev = torch.cuda.device_count()
n, rp, ci, va, low = synth.generate(name, scale)
x = torch.from_numpy(synth.make_x(n)).cuda()
y = torch.empty(n, dtype=torch.float64, device="cuda")
lib = _lib.load()


def timed(A, iters=100):
    for _ in range(10):
        A.dense_vector_multiply(y, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        A.dense_vector_multiply(y, x)
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters, 5)


W = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=32))
out = {"matrix": name, "scale": scale, "n": n, "nnz_full": int(rp[-1]), "visible_devices": ndev,
       "one_handle_ms": timed(W), "forms": {}}
yref = y.clone()
W.close()
for N in counts:
    res = {}
    A = cfs.SymMatrix(n, rp, ci, va, ngpus=N, options=cfs.make_options(flags=32))
    for mode, tag in ((0, "mirror_x_peer"), (1, "mirror_x_replicated"), (2, "mirror_x_replicated_all_shards")):
        _lib.check(lib.cfs_hip_sym_multi_set_xmode(A._h, mode))
        res[tag] = timed(A)
        assert float((y - yref).abs().max()) < 1e-9
    A.close()
    A = cfs.SymMatrix(n, rp, ci, va, ngpus=N, options=cfs.make_options(flags=32 | cfs.FLAG_SHARD_EXCHANGE))
    res["exchange_reduce_scatter_native"] = timed(A)
    assert float((y - yref).abs().max()) < 1e-9
    A.close()
    res["shards_per_device"] = -(-N // max(1, ndev))
    out["forms"][str(N)] = res
out["note"] = ("rehearsal: shards share a device, peer transport" if ndev < max(counts)
               else "one shard per device")
print(json.dumps(out))
