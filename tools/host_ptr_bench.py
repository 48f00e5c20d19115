"""developer tool: the PCIe-inclusive rate of the drop-in host-pointer path
(cfs_hip_sym_spmv with host x / y): pageable vectors (copied through the handle's
page-locked blocks) and vectors that live in page-locked memory (DMA in place)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cfs_spmv_amd as cfs
from cfs_spmv_amd import synth, _lib
name = sys.argv[1] if len(sys.argv) > 1 else "Flan_1565"
n, rp, ci, va, low = synth.generate(name, 1.0)
lib = _lib.load()
A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=32))
x = synth.make_x(n); y = np.empty(n)
def timeit(f, k=20):
    f(); f()
    t = time.time()
    for _ in range(k): f()
    return (time.time() - t) / k * 1e3
t_page = timeit(lambda: A.dense_vector_multiply_host(y, x))
px, py = C.c_void_p(), C.c_void_p()
_lib.check(lib.cfs_hip_alloc(n * 8, 1, C.byref(px))); _lib.check(lib.cfs_hip_alloc(n * 8, 1, C.byref(py)))
np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_double)), shape=(n,))[:] = x
t_pin = timeit(lambda: _lib.check(lib.cfs_hip_sym_spmv(A._h, py, px)))
nnz = int(rp[-1])
print(f"{name}: host-pointer SpMV, pageable x/y {t_page:.3f} ms = {2*nnz/t_page/1e6:.0f} GFLOP/s; "
      f"page-locked x/y {t_pin:.3f} ms = {2*nnz/t_pin/1e6:.0f} GFLOP/s (2 x {n*8/1e6:.1f} MB over PCIe per call)")
