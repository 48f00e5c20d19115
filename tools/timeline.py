"""developer tool: per-workgroup phase timeline of the tile kernel"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cfs_spmv_amd as cfs
from cfs_spmv_amd import synth, _lib
name = sys.argv[1] if len(sys.argv) > 1 else "Flan_1565"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cfgs = sys.argv[3:] or ["2560,256,0,0"]
n, rp, ci, va, low = synth.generate(name, scale)
x = torch.from_numpy(synth.make_x(n)).cuda(); y = torch.empty(n, dtype=torch.float64, device="cuda")
for cfg in cfgs:
    sl, bl, mt, fl = (int(v) for v in cfg.split(","))
    A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(sl, mt, bl, fl))
    buf = np.zeros(8 * 8192, dtype=np.uint64); ng = C.c_int()
    _lib.check(_lib.load().cfs_hip_sym_debug_timeline(A._h, y.data_ptr(), x.data_ptr(), buf.ctypes.data, buf.size, C.byref(ng)))
    t = buf[:ng.value * 8].reshape(-1, 8).astype(np.int64)
    ends = []
    for rep in range(4):
        b2 = np.zeros_like(buf)
        _lib.check(_lib.load().cfs_hip_sym_debug_timeline(A._h, y.data_ptr(), x.data_ptr(), b2.ctypes.data, b2.size, C.byref(ng)))
        tt = b2[:ng.value * 8].reshape(-1, 8).astype(np.int64)
        ends.append((tt[:, 3] - tt[:, 0].min()) / 100.0)
    ends = np.array(ends)
    print("  repeatability: corr(run0,run1) %.3f corr(run0,run3) %.3f ; per-group std over runs mean %.2f us ; kernel end per run" % (np.corrcoef(ends[0], ends[1])[0,1], np.corrcoef(ends[0], ends[3])[0,1], ends.std(0).mean()), ends.max(1))
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    def q(v): return "min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % (v.min(), np.percentile(v,10), np.median(v), np.percentile(v,90), v.max())
    print(cfg, "groups", ng.value, "tiles", A.stats()["ntiles"])
    print("  start      ", q(us[:, 0]))
    print("  x ready    ", q(us[:, 1]))
    print("  slices done", q(us[:, 2]))
    print("  end        ", q(us[:, 3]))
    print("  tile desc   ", q(us[:, 4]))
    print("  slot table  ", q(us[:, 5]))
    print("  x gathered  ", q(us[:, 6]))
    end = us[:, 3]; dur = us[:, 3] - us[:, 1]
    shift = 0
    nper = ng.value // 8
    grp = ((np.arange(ng.value) + shift) % 8) * nper + np.arange(ng.value) // 8
    for k in range(8):
        sel = grp // nper == k
        print("   row-range  %d: end mean %.1f std %.1f" % (k, end[sel].mean(), end[sel].std()))
    for k in range(8):
        sel = np.arange(ng.value) % 8 == k
        print("   xcd-label %d: end mean %.1f std %.1f | stream-phase mean %.1f std %.1f" % (k, end[sel].mean(), end[sel].std(), dur[sel].mean(), dur[sel].std()))
    # same CU slot? blocks b, b+8*32.. unknown; print autocorrelation by rank order
    q = max(1, ng.value // 4)
    for k in range(4):
        sel = (np.arange(ng.value) // q) == k
        if sel.any(): print("   dispatch quartile %d: x-ready mean %.1f | end mean %.1f std %.1f | stream-phase mean %.1f" % (k, us[sel,1].mean(), end[sel].mean(), end[sel].std(), dur[sel].mean()))
    order = np.argsort(end)
    print("   slowest 10 blocks:", order[-10:], " fastest 10:", order[:10])
    A.close()
