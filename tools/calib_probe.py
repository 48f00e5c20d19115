"""developer tool: feedback calibration of the per-group cost shares from the tile kernel's
per-workgroup timeline (does the launch get shorter when slow XCDs / second-wave workgroups get less work?)"""
import sys, os, ctypes as C, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cfs_spmv_amd as cfs
from cfs_spmv_amd import synth, _lib
name = sys.argv[1] if len(sys.argv) > 1 else "Flan_1565"
mode = sys.argv[2] if len(sys.argv) > 2 else "xw"      # xw = (xcd, wave) model, g = per group
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = sys.argv[4] if len(sys.argv) > 4 else "0,0,0,32"
alpha = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
dt = np.float32 if (len(sys.argv) > 6 and sys.argv[6] == "f32") else np.float64
n, rp, ci, va, low = synth.generate(name, 1.0)
va = va.astype(dt)
tdt = torch.float32 if dt == np.float32 else torch.float64
x = torch.from_numpy(synth.make_x(n).astype(dt)).cuda(); y = torch.empty(n, dtype=tdt, device="cuda")
sl, bl, mt, fl = (int(v) for v in cfg.split(","))
share = None
path = os.path.join(tempfile.gettempdir(), "cfs_share.txt")
for it in range(iters + 1):
    if share is not None:
        np.savetxt(path, share)
        os.environ["CFS_HIP_GROUP_SHARE_FILE"] = path
    A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(sl, mt, bl, fl))
    st = A.stats()
    G = st["ngroups"]; nper = G // 8
    buf = np.zeros(8 * 8192, dtype=np.uint64); ng = C.c_int()
    D = []; E = []; K = []
    for rep in range(6):
        _lib.check(_lib.load().cfs_hip_sym_debug_timeline(A._h, y.data_ptr(), x.data_ptr(), buf.ctypes.data, buf.size, C.byref(ng)))
        t = buf[:ng.value * 8].reshape(-1, 8).astype(np.int64)
        if rep < 2: continue
        t0 = t[:, 0].min()
        D.append((t[:, 3] - t[:, 1]) / 100.0); E.append((t[:, 3] - t0) / 100.0); K.append((t[:, 3].max() - t0) / 100.0)
    D = np.mean(D, 0); E = np.mean(E, 0)
    # whole SpMV with events
    for _ in range(30): A.dense_vector_multiply(y, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): A.dense_vector_multiply(y, x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 300
    b = np.arange(G)
    xcd = b & 7; wave = (b >= G // 2).astype(int) if st["block_threads"] == 512 else np.zeros(G, int)
    print("iter %d: groups %d tiles %d halo %d | kernel end %.1f us (runs %s) | WG end mean %.1f min %.1f max %.1f std %.2f | SpMV %.2f us"
          % (it, G, st["ntiles"], st["halo_slots"], np.mean(K), np.round(K, 1), E.mean(), E.min(), E.max(), E.std(), ms * 1e3), flush=True)
    print("   xcd end  ", " ".join("%.1f" % E[xcd == k].mean() for k in range(8)), "| wave end", " ".join("%.1f" % E[wave == w].mean() for w in range(wave.max() + 1)))
    # measure of speed: the end time a group would need per unit of its share
    if share is None:
        share = np.ones(G)
    gshare_of_block = share[(b & 7) * nper + (b >> 3)]
    if mode == "xw":
        corr_b = np.ones(G)
        for k in range(8):
            for w in range(wave.max() + 1):
                sel = (xcd == k) & (wave == w)
                corr_b[sel] = E.mean() / E[sel].mean()
    else:
        corr_b = E.mean() / E
    newb = gshare_of_block * corr_b ** alpha
    newshare = np.empty(G)
    newshare[(b & 7) * nper + (b >> 3)] = newb
    share = newshare / newshare.mean()
    A.close()
# ---- what are the outliers of the last schedule? ----
os.environ["CFS_HIP_GROUP_SHARE_FILE"] = path
A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(sl, mt, bl, fl))
G = A.stats()["ngroups"]; nper = G // 8
ends = []
for rep in range(8):
    _lib.check(_lib.load().cfs_hip_sym_debug_timeline(A._h, y.data_ptr(), x.data_ptr(), buf.ctypes.data, buf.size, C.byref(ng)))
    t = buf[:ng.value * 8].reshape(-1, 8).astype(np.int64)
    ends.append((t[:, [1, 2, 3]] - t[:, 0].min()) / 100.0)
ends = np.array(ends)[2:]
feat = np.zeros(G * 10, dtype=np.int64)
_lib.check(_lib.load().cfs_hip_sym_debug_group_features(A._h, feat.ctypes.data, feat.size, C.byref(ng)))
feat = feat.reshape(-1, 10)
b = np.arange(G); grp = (b & 7) * nper + (b >> 3)
F = feat[grp]
E = ends[:, :, 2].mean(0)
print("per-block end: std over runs (mean) %.2f ; corr run-to-run %.2f" % (ends[:, :, 2].std(0).mean(), np.corrcoef(ends[0, :, 2], ends[-1, :, 2])[0, 1]))
order = np.argsort(E)
med = np.median(F, 0)
print("median features [tiles rows vrows slices rounds vals slots coo halo nslots]:", med)
for bb in order[-16:][::-1]:
    print("  block %3d xcd %d wave %d end %.1f (runs %s) xready %.1f | feat/med %s tiles %d" % (bb, bb & 7, int(bb >= G // 2), E[bb], np.round(ends[:, bb, 2], 0), ends[:, bb, 0].mean(), np.round(F[bb, 1:] / np.maximum(med[1:], 1), 2), F[bb, 0]))
print(" fastest:")
for bb in order[:6]:
    print("  block %3d xcd %d wave %d end %.1f xready %.1f | feat/med %s tiles %d" % (bb, bb & 7, int(bb >= G // 2), E[bb], ends[:, bb, 0].mean(), np.round(F[bb, 1:] / np.maximum(med[1:], 1), 2), F[bb, 0]))
X = np.column_stack([F[:, [0, 1, 3, 4, 5, 6, 7, 8]].astype(float), (b >= G // 2).astype(float)] + [(b & 7) == k for k in range(8)])
coef, *_ = np.linalg.lstsq(X, E, rcond=None)
res = E - X @ coef
print(" fit end ~ [tiles rows slices rounds vals slots coo halo wave xcd0..7]:", np.array2string(coef, precision=6), "resid std %.2f (end std %.2f)" % (res.std(), E.std()))
