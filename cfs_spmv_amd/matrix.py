"""Host-side mirror of the reference's operator surface for the hot path, over
the C ABI (include/cfs_hip.h).  Names and argument meaning follow

    SparseMatrix<I,V>::create / nrows / ncols / nnz / symmetric / size / tune /
    dense_vector_multiply     include/matrix/sparse_matrix.hpp:23-41
    SpDMV<I,V>(A, Tuning), operator()(y, M, x, N)
                              include/kernel/sparse_kernel.hpp:17-27

so the parity tests read like test/test_spmv_mmf.cpp.  torch is used only for
device memory and streams; nothing here computes an SpMV on the host."""
import ctypes as C
import enum

import numpy as np

from . import _lib


class Format(enum.IntEnum):  # include/utils/platform.hpp:23
    none = 0
    csr = 1
    sss = 2
    hyb = 3


class Tuning(enum.IntEnum):  # include/utils/platform.hpp:22
    NONE = 0
    Aggressive = 1


class Kernel(enum.IntEnum):  # include/utils/platform.hpp:21
    SpDMV = 0


def _np_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(t):
    """raw pointer of a torch tensor / numpy array / int"""
    if isinstance(t, int):
        return t
    if isinstance(t, np.ndarray):
        return t.ctypes.data
    return t.data_ptr()


def _stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return s.cuda_stream


FLAG_SHARD_EXCHANGE = 64  # include/cfs_hip.h: CFS_HIP_FLAG_SHARD_EXCHANGE
FLAG_KEEP_VALUE_MAP = 2048  # CFS_HIP_FLAG_KEEP_VALUE_MAP
FLAG_HOST_PLAN = 4096  # CFS_HIP_FLAG_HOST_PLAN: build the schedule with the host builder
DIGEST_WORDS = 28  # CFS_HIP_DIGEST_WORDS
DIGEST_NAMES = ["tiles", "gfirst", "group_range", "slot_col", "rowinfo", "diag", "slice_meta", "leadlane",
                "vals", "slots", "cvals", "crows", "ccols", "fold_rec", "fold_idx", "val_map", "cval_map",
                "diag_map", "window", "slot_exp", "send_ptr", "send_idx", "fvals", "frows", "fcols", "fval_map",
                "far_entries", "device_built"]


def make_options(max_slots=0, max_tile_nnz=0, block_threads=0, flags=0):
    return _lib.Options(max_slots, max_tile_nnz, block_threads, flags)


def balanced_splits(n, rowptr, colind, nranks):
    rowptr, colind = _np_i32(rowptr), _np_i32(colind)
    out = np.zeros(nranks + 1, dtype=np.int32)
    _lib.check(_lib.load().cfs_hip_sym_balanced_splits(
        n, rowptr.ctypes.data, colind.ctypes.data, nranks, out.ctypes.data))
    return out


def plan_check(n, rowptr, colind, values, nranks=1, rank=0, row_splits=None, options=None):
    """host-only structural self-check of the tile schedule (no GPU needed)"""
    lib = _lib.load()
    rowptr, colind = _np_i32(rowptr), _np_i32(colind)
    values = np.ascontiguousarray(values)
    suf = "f64" if values.dtype == np.float64 else "f32"
    rs = _np_i32(row_splits) if row_splits is not None else None
    rep = _lib.PlanReport()
    _lib.check(getattr(lib, "cfs_hip_sym_plan_check_" + suf)(
        n, rowptr.ctypes.data, colind.ctypes.data, values.ctypes.data, nranks, rank,
        rs.ctypes.data if rs is not None else None,
        C.byref(options) if options is not None else None, C.byref(rep)))
    return rep.asdict()


def plan_send_info(n, rowptr, colind, values, nranks, rank, row_splits, options=None):
    """host-only (send_counts, send_rows) of one shard in the EXCHANGE form
    (CFS_HIP_FLAG_SHARD_EXCHANGE) -- for the CPU exchange tests"""
    lib = _lib.load()
    if options is None:
        options = make_options(flags=FLAG_SHARD_EXCHANGE)
    else:
        options = _lib.Options(options.max_slots, options.max_tile_nnz, options.block_threads,
                               options.flags | FLAG_SHARD_EXCHANGE)
    rowptr, colind = _np_i32(rowptr), _np_i32(colind)
    values = np.ascontiguousarray(values, dtype=np.float64)
    rs = _np_i32(row_splits)
    counts = np.zeros(nranks, dtype=np.int32)
    nr = C.c_int()
    optp = C.byref(options) if options is not None else None
    _lib.check(lib.cfs_hip_sym_plan_send_info_f64(
        n, rowptr.ctypes.data, colind.ctypes.data, values.ctypes.data, nranks, rank,
        rs.ctypes.data, optp, counts.ctypes.data, None, 0, C.byref(nr)))
    rows = np.zeros(nr.value, dtype=np.int32)
    if nr.value:
        _lib.check(lib.cfs_hip_sym_plan_send_info_f64(
            n, rowptr.ctypes.data, colind.ctypes.data, values.ctypes.data, nranks, rank,
            rs.ctypes.data, optp, counts.ctypes.data, rows.ctypes.data, nr.value,
            C.byref(nr)))
    return counts, rows


class SymMatrix:
    """A symmetric matrix tuned for the MI355X tile kernel (Format::sss).

    Built from the FULL CSR exactly as CSRMatrix holds it before tune()
    (csr_matrix.tpp:74-107).  `row_splits`/`rank` build one 1-D row block of a
    sharded matrix."""

    def __init__(self, n, rowptr, colind, values, options=None, row_splits=None, rank=0,
                 ngpus=1, devices=None):
        lib = _lib.load()
        rowptr, colind = _np_i32(rowptr), _np_i32(colind)
        values = np.ascontiguousarray(values)
        if values.dtype not in (np.float64, np.float32):
            raise TypeError("values must be float64 or float32 (src/csr.cpp:10-11)")
        self.dtype = values.dtype
        self.n = int(n)
        suf = "f64" if self.dtype == np.float64 else "f32"
        self._h = C.c_void_p()
        optp = C.byref(options) if options is not None else None
        if ngpus > 1:
            # one host thread, ngpus shards (cfs_hip_sym_create_multi_*: what the C++
            # surface builds for CFS_NUM_GPUS); shards may share devices
            dv = _np_i32(devices) if devices is not None else None
            _lib.check(getattr(lib, "cfs_hip_sym_create_multi_" + suf)(
                n, rowptr.ctypes.data, colind.ctypes.data, values.ctypes.data, int(ngpus),
                dv.ctypes.data if dv is not None else None, optp, C.byref(self._h)))
            self.nranks, self.rank = 1, 0
        elif row_splits is None:
            _lib.check(getattr(lib, "cfs_hip_sym_create_" + suf)(
                n, rowptr.ctypes.data, colind.ctypes.data, values.ctypes.data, optp,
                C.byref(self._h)))
            self.nranks, self.rank = 1, 0
        else:
            rs = _np_i32(row_splits)
            self.nranks, self.rank = len(rs) - 1, int(rank)
            _lib.check(getattr(lib, "cfs_hip_sym_create_shard_" + suf)(
                n, rowptr.ctypes.data, colind.ctypes.data, values.ctypes.data, self.nranks,
                self.rank, rs.ctypes.data, optp, C.byref(self._h)))
        self._tuned = True
        st = self.stats()
        self.row_begin, self.row_end = st["row_begin"], st["row_end"]

    # -- SparseMatrix surface (sparse_matrix.hpp:25-33) --
    def nrows(self):
        return self.n

    def ncols(self):
        return self.n

    def nnz(self):
        return self.stats()["nnz_full"]

    def symmetric(self):
        return True

    def size(self):
        return self.stats()["device_bytes"]

    def tune(self, kernel=Kernel.SpDMV, tuning=Tuning.Aggressive):
        return True  # the schedule is built at construction

    def digest(self):
        """developer / test: digests of the schedule's device arrays (cfs_hip_sym_debug_digest)"""
        w = (C.c_ulonglong * DIGEST_WORDS)()
        _lib.check(_lib.load().cfs_hip_sym_debug_digest(self._h, w, DIGEST_WORDS))
        return dict(zip(DIGEST_NAMES, [int(v) for v in w]))

    def plan_note(self):
        """why the device builder handed the schedule to the host builder ('' = it built it)"""
        buf = C.create_string_buffer(256)
        _lib.check(_lib.load().cfs_hip_sym_debug_plan_note(self._h, buf, 256))
        return buf.value.decode(errors="replace")

    def stats(self):
        st = _lib.SymStats()
        _lib.check(_lib.load().cfs_hip_sym_get_stats(self._h, C.byref(st)))
        return st.asdict()

    def dense_vector_multiply(self, y, x, stream=None):
        """y <- A x on device tensors (fully overwrites y), enqueued on torch's
        current stream (or `stream`)."""
        _lib.check(_lib.load().cfs_hip_sym_spmv_async(
            self._h, _ptr(y), _ptr(x), _stream_ptr(stream)))

    def dense_vector_multiply_host(self, y, x):
        """host numpy arrays: the slow staged drop-in path of cfs_hip_sym_spmv"""
        _lib.check(_lib.load().cfs_hip_sym_spmv(self._h, _ptr(y), _ptr(x)))

    def cg(self, u, b, tol=1e-10, maxiter=1000, check_every=8, stream=None):
        """conjugate gradients inside the library (cfs_hip_sym_cg): u (device tensor) holds the
        first guess and receives the solution of A u = b; no host round trip inside the loop.
        Returns (iterations, ||b - A u|| / ||b||)."""
        it, res = C.c_int(), C.c_double()
        _lib.check(_lib.load().cfs_hip_sym_cg(self._h, _ptr(u), _ptr(b), float(tol), int(maxiter), int(check_every),
                                              C.byref(it), C.byref(res), _stream_ptr(stream)))
        return it.value, res.value

    # -- sharded operation --
    def send_counts(self):
        out = np.zeros(self.nranks, dtype=np.int32)
        _lib.check(_lib.load().cfs_hip_sym_shard_send_counts(self._h, out.ctypes.data))
        return out

    def send_rows(self):
        out = np.zeros(int(self.stats()["remote_vals"]), dtype=np.int32)
        if out.size:
            _lib.check(_lib.load().cfs_hip_sym_shard_send_rows(self._h, out.ctypes.data))
        return out

    def set_recv(self, recv_rows):
        recv_rows = _np_i32(recv_rows)
        _lib.check(_lib.load().cfs_hip_sym_shard_set_recv(
            self._h, recv_rows.size, recv_rows.ctypes.data if recv_rows.size else None))

    def spmv_local(self, y_block, x, send_buf, stream=None):
        _lib.check(_lib.load().cfs_hip_sym_spmv_local_async(
            self._h, _ptr(y_block), _ptr(x), _ptr(send_buf) if send_buf is not None else None,
            _stream_ptr(stream)))

    def spmv_phases(self, y_block, x, send_buf, phases, stream=None):
        """enqueue only the selected launches (1 = tile kernel, 2 = halo fold, 4 = pack)"""
        _lib.check(_lib.load().cfs_hip_sym_spmv_phases_async(
            self._h, _ptr(y_block), _ptr(x), _ptr(send_buf) if send_buf is not None else None,
            int(phases), _stream_ptr(stream)))

    def recv_fold(self, y_block, recv_buf, stream=None):
        _lib.check(_lib.load().cfs_hip_sym_recv_fold_async(
            self._h, _ptr(y_block), _ptr(recv_buf) if recv_buf is not None else None,
            _stream_ptr(stream)))

    def update_values(self, values):
        """new values, same sparsity pattern (numpy array or device tensor in the order of the
        CSR the matrix was created from); needs FLAG_KEEP_VALUE_MAP at construction"""
        suf = "f64" if self.dtype == np.float64 else "f32"
        if isinstance(values, np.ndarray):
            values = np.ascontiguousarray(values, dtype=self.dtype)
            ptr, cnt = values.ctypes.data, values.size
        else:
            ptr, cnt = values.data_ptr(), values.numel()
        _lib.check(getattr(_lib.load(), "cfs_hip_sym_update_values_" + suf)(self._h, ptr, cnt))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.load().cfs_hip_sym_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CsrMatrix:
    """General CSR on the GPU (Format::csr; cpu_mv's role, csr_matrix.tpp:2683-2704)."""

    def __init__(self, nrows, ncols, rowptr, colind, values):
        lib = _lib.load()
        rowptr, colind = _np_i32(rowptr), _np_i32(colind)
        values = np.ascontiguousarray(values)
        self.dtype = values.dtype
        self._nrows, self._ncols, self._nnz = int(nrows), int(ncols), int(rowptr[-1])
        suf = "f64" if self.dtype == np.float64 else "f32"
        self._h = C.c_void_p()
        _lib.check(getattr(lib, "cfs_hip_csr_create_" + suf)(
            nrows, ncols, rowptr.ctypes.data, colind.ctypes.data, values.ctypes.data,
            C.byref(self._h)))

    def nrows(self):
        return self._nrows

    def ncols(self):
        return self._ncols

    def nnz(self):
        return self._nnz

    def symmetric(self):
        return False

    def tune(self, kernel=Kernel.SpDMV, tuning=Tuning.Aggressive):
        return True

    def dense_vector_multiply(self, y, x, stream=None):
        _lib.check(_lib.load().cfs_hip_csr_spmv_async(
            self._h, _ptr(y), _ptr(x), _stream_ptr(stream)))

    def dense_vector_multiply_host(self, y, x):
        _lib.check(_lib.load().cfs_hip_csr_spmv(self._h, _ptr(y), _ptr(x)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.load().cfs_hip_csr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SpDMV:
    """include/kernel/sparse_kernel.hpp:17-27: ctor tunes, call multiplies."""

    def __init__(self, A, tuning=Tuning.Aggressive):
        self.A = A
        A.tune(Kernel.SpDMV, tuning)

    def __call__(self, y, M, x, N):
        assert self.A.nrows() == M  # sparse_kernel.tpp:23-24
        assert self.A.ncols() == N
        if isinstance(y, np.ndarray):
            self.A.dense_vector_multiply_host(y, x)
        else:
            self.A.dense_vector_multiply(y, x)
