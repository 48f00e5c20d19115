"""developer probe: what CPU allowance does this box give us, and how does the oracle's
conflict-free SpMV scale with the thread count?  (decides cpu_baseline's thread count)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
          "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "-", type(e).__name__)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
print("loadavg", open("/proc/loadavg").read().strip())
import numpy as np
from cfs_spmv_amd import synth
from oracle import oracle
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
n, rp, ci, va, low = synth.generate("Flan_1565", scale)
x = synth.make_x(n)
for T in [int(a) for a in sys.argv[2:]] or [8, 16, 32, 64, 96]:
    t0 = time.time()
    o = oracle.SymOracle(n, rp, ci, va, T)
    pre = time.time() - t0
    y = o.spmv(x)
    for _ in range(5):
        o.spmv(x, y)
    t0 = time.time()
    for _ in range(20):
        o.spmv(x, y)
    dt = (time.time() - t0) / 20
    print(f"T={T} preproc {pre:.2f}s  {dt*1e3:.3f} ms/SpMV  {2.0*int(rp[-1])/dt/1e9:.1f} GFLOP/s", flush=True)
    o.close()
