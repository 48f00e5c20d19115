"""Matrix-Market files through the C++ surface's parallel reader (src/mmf.cpp ->
build/libsparse.so, C entry cfs_mmf_load_csr_f64): the same arrays CSRMatrix holds
before tune() -- full CSR, symmetric files expanded (reference: include/io/mmf.hpp,
csr_matrix.tpp:8-111).  CFS_MTX_CACHE_DIR enables the reader's binary side files."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_mtx(path):
    """-> dict(nrows, ncols, nnz, symmetric, rowptr int32, colind int32, values float64)"""
    so = os.path.join(ROOT, "build", "libsparse.so")
    if not os.path.exists(so):
        raise FileNotFoundError(f"{so} is missing: run `make DP=1` (or __graft_entry__.build())")
    lib = C.CDLL(so)
    n, m, sym, nnz = C.c_int(), C.c_int(), C.c_int(), C.c_long()
    rp, ci, va = C.c_void_p(), C.c_void_p(), C.c_void_p()
    err = C.create_string_buffer(256)
    rc = lib.cfs_mmf_load_csr_f64(os.fsencode(path), C.byref(n), C.byref(m), C.byref(nnz),
                                  C.byref(sym), C.byref(rp), C.byref(ci), C.byref(va), err, 256)
    if rc != 0:
        raise ValueError(f"{path}: {err.value.decode(errors='replace')}")
    lib.cfs_mmf_free.argtypes = [C.c_void_p]

    def take(p, cnt, dt):
        nb = cnt * np.dtype(dt).itemsize
        a = (np.frombuffer((C.c_char * nb).from_address(p.value), dtype=dt, count=cnt).copy()
             if cnt else np.zeros(0, dt))
        lib.cfs_mmf_free(p)
        return a
    return dict(nrows=n.value, ncols=m.value, nnz=nnz.value, symmetric=bool(sym.value),
                rowptr=take(rp, n.value + 1, np.int32), colind=take(ci, nnz.value, np.int32),
                values=take(va, nnz.value, np.float64))


def find_real_matrix(name):
    """the real SuiteSparse file for a config, if the user provides it:
    $CFS_MTX_DIR/<name>.mtx (the stand-ins are only used without it)"""
    d = os.environ.get("CFS_MTX_DIR")
    if not d:
        return None
    p = os.path.join(d, name + ".mtx")
    return p if os.path.exists(p) else None
