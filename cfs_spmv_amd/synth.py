"""Synthetic stand-ins for the SuiteSparse matrices BASELINE.json names
(SURVEY.md section 8d).  Workload generation only (csrc/cfs_synth.c)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NAMES = ("pdb1HYS", "pwtk", "ldoor", "Flan_1565", "Queen_4147")
# non-regular stand-ins (SURVEY 7 "structure-faithful synthetic generators"): a random 3-D
# point cloud, 21 nearest neighbours, 3 dof per node; random / breadth-first numbering
UNSTRUCTURED = ("unstruct", "unstruct_bfs")
# tetmesh: the same kind of graph with 1..4 unknowns per node whose rows do not repeat each
# other's columns; powerlaw: heavy-tailed row lengths, hub columns
NONREGULAR = ("tetmesh", "powerlaw")


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "libcfs_synth.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run cfs_spmv_amd/build.py")
        lib = C.CDLL(path)
        lib.cfs_synth_generate.argtypes = [
            C.c_char_p, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_long),
            C.POINTER(C.c_long), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
            C.POINTER(C.c_void_p)]
        lib.cfs_synth_free.argtypes = [C.c_void_p]
        lib.cfs_synth_x.argtypes = [C.c_int, C.c_uint64, C.c_void_p]
        lib.cfs_synth_write_mtx.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_int]
        _LIB = lib
    return _LIB


def _take(ptr, count, dtype):
    """copy a malloc'ed C array into a numpy array and free it"""
    nbytes = count * np.dtype(dtype).itemsize
    buf = (C.c_char * nbytes).from_address(ptr.value)
    out = np.frombuffer(buf, dtype=dtype, count=count).copy()
    _lib().cfs_synth_free(ptr)
    return out


def generate(name, scale=1.0):
    """-> (n, rowptr[int32 n+1], colind[int32 nnz_full], values[float64 nnz_full], nnz_low)
    full (both triangles) CSR, rows and columns ascending, 0-based."""
    lib = _lib()
    n, nnz, low = C.c_int(), C.c_long(), C.c_long()
    rp, ci, va = C.c_void_p(), C.c_void_p(), C.c_void_p()
    rc = lib.cfs_synth_generate(name.encode(), float(scale), C.byref(n), C.byref(nnz),
                                C.byref(low), C.byref(rp), C.byref(ci), C.byref(va))
    if rc != 0:
        raise ValueError(f"cfs_synth_generate({name!r}) failed: {rc}")
    rowptr = _take(rp, n.value + 1, np.int32)
    colind = _take(ci, nnz.value, np.int32)
    values = _take(va, nnz.value, np.float64)
    return n.value, rowptr, colind, values, low.value


def make_x(n, seed=42, dtype=np.float64):
    """x_i = 0.01 + 0.41 u_i (64-bit LCG), the benchmark input vector"""
    x = np.empty(n, dtype=np.float64)
    _lib().cfs_synth_x(n, seed, x.ctypes.data)
    return x.astype(dtype, copy=False)


def write_mtx(path, n, rowptr, colind, values, general=False):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colind = np.ascontiguousarray(colind, dtype=np.int32)
    values = np.ascontiguousarray(values, dtype=np.float64)
    rc = _lib().cfs_synth_write_mtx(os.fsencode(path), n, rowptr.ctypes.data,
                                    colind.ctypes.data, values.ctypes.data, int(general))
    if rc != 0:
        raise OSError(f"cannot write {path}")


def random_symmetric(n, avg_lower, seed, band=None, dtype=np.float64, full_diag=True):
    """small random symmetric test matrix (numpy only) -> full CSR"""
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for i in range(n):
        k = int(rng.poisson(avg_lower)) if i > 0 else 0
        if k:
            lo = 0 if band is None else max(0, i - band)
            c = np.unique(rng.integers(lo, i, size=k))
            rows.append(np.full(c.size, i))
            cols.append(c)
    if rows:
        r = np.concatenate(rows)
        c = np.concatenate(cols)
    else:
        r = np.zeros(0, dtype=np.int64)
        c = np.zeros(0, dtype=np.int64)
    v = rng.uniform(-1.0, 1.0, size=r.size)
    d = rng.uniform(1.0, 2.0, size=n) + 0.0
    import scipy.sparse as sp
    L = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr()
    A = L + L.T
    if full_diag:
        A = A + sp.diags(d)
    A = A.tocsr()
    A.sort_indices()
    return (n, A.indptr.astype(np.int32), A.indices.astype(np.int32),
            A.data.astype(dtype))
