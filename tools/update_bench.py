"""developer tool: cost of refreshing the values of an existing schedule on the device
(cfs_hip_sym_update_values_*) against a second tune().  usage: update_bench.py <matrix> [scale]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cfs_spmv_amd as cfs
from cfs_spmv_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "Flan_1565"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
n, rp, ci, va, low = synth.generate(name, scale)
t = time.time()
A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=32 | cfs.FLAG_KEEP_VALUE_MAP))
t_tune = time.time() - t
va2 = va * 0.5
vd = torch.from_numpy(va2).cuda()
torch.cuda.synchronize()
for kind, arg in (("host pointer", va2), ("device pointer", vd)):
    A.update_values(arg)
    t = time.time()
    for _ in range(5):
        A.update_values(arg)
    print(f"{name}: tune() {t_tune:.2f} s (Tuning::None, with value map); update_values from a {kind}: "
          f"{(time.time() - t) / 5 * 1e3:.2f} ms; device bytes {A.stats()['device_bytes'] / 1e6:.0f} MB", flush=True)
