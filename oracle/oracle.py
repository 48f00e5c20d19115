"""ctypes wrapper of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker / reported CPU baseline.  See
oracle/cfs_oracle.h for the pinning status of each part."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcInfo(C.Structure):
    _fields_ = [("nthreads", C.c_int), ("ncolors", C.c_int), ("nranges", C.c_int),
                ("nnz_low", C.c_int), ("nnz_diag", C.c_int), ("nvertices", C.c_int),
                ("nedges", C.c_int), ("size_bytes", C.c_size_t)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle`")
        L = C.CDLL(path)
        vp = C.c_void_p
        L.orc_last_error.restype = C.c_char_p
        L.orc_free.argtypes = [vp]
        for suf in ("f64", "f32"):
            getattr(L, "orc_mmf_load_" + suf).argtypes = [
                C.c_char_p] + [C.POINTER(C.c_int)] * 4 + [C.POINTER(vp)] * 3
            f = getattr(L, "orc_sym_build_" + suf)
            f.argtypes = [C.c_int, vp, vp, vp, C.c_int]
            f.restype = vp
            getattr(L, "orc_sym_spmv_" + suf).argtypes = [vp, vp, vp]
            getattr(L, "orc_csr_spmv_" + suf).argtypes = [C.c_int, vp, vp, vp, C.c_int, vp, vp, vp]
            getattr(L, "orc_csr_spmv_ld_" + suf).argtypes = [C.c_int, vp, vp, vp, vp, vp, vp]
        L.orc_sym_free.argtypes = [vp]
        L.orc_sym_info.argtypes = [vp, C.POINTER(OrcInfo)]
        L.orc_sym_colors.argtypes = [vp, vp]
        L.orc_sym_partition.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                        C.POINTER(vp)]
        L.orc_partition_by_nrows.argtypes = [C.c_int, C.c_int, vp]
        L.orc_partition_by_nnz.argtypes = [C.c_int, vp, C.c_int, vp]
        L.orc_is_equal_f64.argtypes = [C.c_double, C.c_double]
        L.orc_is_equal_f32.argtypes = [C.c_float, C.c_float]
        _LIB = L
    return _LIB


def _suf(dtype):
    return "f64" if np.dtype(dtype) == np.float64 else "f32"


def last_error():
    return lib().orc_last_error().decode()


def _copy(ptr, count, dtype):
    nbytes = count * np.dtype(dtype).itemsize
    out = np.frombuffer((C.c_char * nbytes).from_address(ptr.value), dtype=dtype,
                        count=count).copy() if count else np.zeros(0, dtype)
    lib().orc_free(ptr)
    return out


def mmf_load(path, dtype=np.float64):
    """MMF file -> dict(nrows, ncols, nnz, symmetric, rowptr, colind, values);
    raises ValueError(code, message) where the reference would print + exit(1)"""
    L = lib()
    n, m, nnz, sym = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rp, ci, va = C.c_void_p(), C.c_void_p(), C.c_void_p()
    rc = getattr(L, "orc_mmf_load_" + _suf(dtype))(
        os.fsencode(path), C.byref(n), C.byref(m), C.byref(nnz), C.byref(sym),
        C.byref(rp), C.byref(ci), C.byref(va))
    if rc != 0:
        raise ValueError(rc, last_error())
    return dict(nrows=n.value, ncols=m.value, nnz=nnz.value, symmetric=bool(sym.value),
                rowptr=_copy(rp, n.value + 1, np.int32), colind=_copy(ci, nnz.value, np.int32),
                values=_copy(va, nnz.value, dtype))


class SymOracle:
    """tune() + dense_vector_multiply() of the reference's SSS path on T threads"""

    def __init__(self, n, rowptr, colind, values, nthreads=1):
        self.dtype = values.dtype
        self.n = n
        self._keep = (np.ascontiguousarray(rowptr, np.int32),
                      np.ascontiguousarray(colind, np.int32), np.ascontiguousarray(values))
        f = getattr(lib(), "orc_sym_build_" + _suf(self.dtype))
        self.h = f(n, self._keep[0].ctypes.data, self._keep[1].ctypes.data,
                   self._keep[2].ctypes.data, nthreads)
        if not self.h:
            raise ValueError(last_error())
        self.nthreads = nthreads

    def spmv(self, x, y=None):
        x = np.ascontiguousarray(x, self.dtype)
        if y is None:
            # garbage-filled on purpose: the kernel must fully overwrite y
            y = np.full(self.n, 123.456, dtype=self.dtype)
        getattr(lib(), "orc_sym_spmv_" + _suf(self.dtype))(self.h, y.ctypes.data, x.ctypes.data)
        return y

    def info(self):
        o = OrcInfo()
        lib().orc_sym_info(self.h, C.byref(o))
        return {k: getattr(o, k) for k, _ in OrcInfo._fields_}

    def colors(self):
        nv = self.info()["nvertices"]
        out = np.zeros(nv, np.int32)
        if lib().orc_sym_colors(self.h, out.ctypes.data) != 0:
            raise ValueError(last_error())
        return out

    def partition(self, t):
        ro, nr = C.c_int(), C.c_int()
        rp, ci, va, dg = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        if lib().orc_sym_partition(self.h, t, C.byref(ro), C.byref(nr), C.byref(rp),
                                   C.byref(ci), C.byref(va), C.byref(dg)) != 0:
            raise ValueError(last_error())

        def view(p, cnt, dt):
            if cnt == 0:
                return np.zeros(0, dt)
            nb = cnt * np.dtype(dt).itemsize
            return np.frombuffer((C.c_char * nb).from_address(p.value), dtype=dt, count=cnt).copy()
        rowptr = view(rp, nr.value + 1, np.int32)
        nnz = int(rowptr[-1]) if nr.value else 0
        return dict(row_offset=ro.value, nrows=nr.value, rowptr=rowptr,
                    colind=view(ci, nnz, np.int32), values=view(va, nnz, self.dtype),
                    diagonal=view(dg, nr.value, self.dtype))

    def close(self):
        if self.h:
            lib().orc_sym_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def partition_by_nrows(n, T):
    out = np.zeros(T + 1, np.int32)
    if lib().orc_partition_by_nrows(n, T, out.ctypes.data) != 0:
        raise ValueError(last_error())
    return out


def partition_by_nnz(n, rowptr, T):
    rowptr = np.ascontiguousarray(rowptr, np.int32)
    out = np.zeros(T + 1, np.int32)
    if lib().orc_partition_by_nnz(n, rowptr.ctypes.data, T, out.ctypes.data) != 0:
        raise ValueError(last_error())
    return out


def csr_spmv(n, rowptr, colind, values, x, nthreads=1, row_split=None):
    """cpu_mv_serial / cpu_mv: the ground truth of the reference's own test"""
    rowptr = np.ascontiguousarray(rowptr, np.int32)
    colind = np.ascontiguousarray(colind, np.int32)
    values = np.ascontiguousarray(values)
    x = np.ascontiguousarray(x, values.dtype)
    y = np.full(n, -7.0, dtype=values.dtype)
    rs = None
    if row_split is not None:
        rs = np.ascontiguousarray(row_split, np.int32)
    getattr(lib(), "orc_csr_spmv_" + _suf(values.dtype))(
        n, rowptr.ctypes.data, colind.ctypes.data, values.ctypes.data, nthreads,
        rs.ctypes.data if rs is not None else None, y.ctypes.data, x.ctypes.data)
    return y


def csr_spmv_ld(n, rowptr, colind, values, x):
    """long-double arbiter: returns (y, sum_j |a_ij x_j|) as float64"""
    rowptr = np.ascontiguousarray(rowptr, np.int32)
    colind = np.ascontiguousarray(colind, np.int32)
    values = np.ascontiguousarray(values)
    x = np.ascontiguousarray(x, values.dtype)
    y = np.zeros(n, np.float64)
    a = np.zeros(n, np.float64)
    getattr(lib(), "orc_csr_spmv_ld_" + _suf(values.dtype))(
        n, rowptr.ctypes.data, colind.ctypes.data, values.ctypes.data, x.ctypes.data,
        y.ctypes.data, a.ctypes.data)
    return y, a


def is_equal(a, b, dtype=np.float64):
    """include/utils/platform.hpp:27-37"""
    if np.dtype(dtype) == np.float64:
        return bool(lib().orc_is_equal_f64(float(a), float(b)))
    return bool(lib().orc_is_equal_f32(float(a), float(b)))
