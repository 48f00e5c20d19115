"""developer tool: per-group timeline of the tile kernel next to what each group has to
do (cfs_hip_sym_debug_group_features) -> a .npz under gpurun_out/ for fitting the cost
model of the row cut offline.  usage: group_fit.py <matrix> <scale> <out.npz> [flags] [nshards rank]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cfs_spmv_amd as cfs
from cfs_spmv_amd import synth, _lib
name, scale, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 32
nsh = int(sys.argv[5]) if len(sys.argv) > 5 else 1
rank = int(sys.argv[6]) if len(sys.argv) > 6 else 0
n, rp, ci, va, low = synth.generate(name, scale)
if os.environ.get("QB_DTYPE") == "f32":
    va = va.astype(np.float32)
x = torch.from_numpy(synth.make_x(n, 42, va.dtype)).cuda()
kw = {}
if nsh > 1:
    kw = dict(row_splits=cfs.balanced_splits(n, rp, ci, nsh), rank=rank)
A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=flags), **kw)
st = A.stats()
y = torch.empty(st["row_end"] - st["row_begin"], dtype=x.dtype, device="cuda")
lib = _lib.load()
ng = C.c_int()
runs = []
for rep in range(6):
    buf = np.zeros(8 * 8192, dtype=np.uint64)
    _lib.check(lib.cfs_hip_sym_debug_timeline(A._h, y.data_ptr(), x.data_ptr(), buf.ctypes.data, buf.size, C.byref(ng)))
    t = buf[:ng.value * 8].reshape(-1, 8).astype(np.int64)
    runs.append((t - t[:, 0].min()) / 100.0)
feat = np.zeros(ng.value * 10, dtype=np.int64)
_lib.check(lib.cfs_hip_sym_debug_group_features(A._h, feat.ctypes.data, feat.size, C.byref(ng)))
feat = feat.reshape(-1, 10)
runs = np.array(runs)            # [rep, block, 8]
G = ng.value
nper = G // 8
b = np.arange(G)
grp = (b % 8) * nper + b // 8    # block -> group
np.savez(out, runs=runs, feat=feat, block_group=grp, stats=np.array([st[k] for k in ("nnz_low", "ntiles", "halo_slots", "lds_bytes", "block_threads")]))
end = runs[1:, :, 3].mean(0); xr = runs[1:, :, 1].mean(0)
print(name, "groups", G, "tiles", st["ntiles"], "end: min %.1f med %.1f max %.1f | x-ready med %.1f" % (end.min(), np.median(end), end.max(), np.median(xr)))
F = feat[grp].astype(float)
X = np.column_stack([F[:, [0, 1, 3, 4, 5, 6, 7, 8]], np.ones(G)])
for target, nm in ((end, "end"), (end - xr, "stream")):
    coef, *_ = np.linalg.lstsq(X, target, rcond=None)
    res = target - X @ coef
    print(" fit", nm, "coef [tiles rows slices rounds vals slots coo halo 1]:", np.array2string(coef, precision=5), "resid std %.2f (target std %.2f)" % (res.std(), target.std()))
