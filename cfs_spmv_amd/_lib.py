"""ctypes binding of the C ABI (include/cfs_hip.h).  No torch types cross the
boundary: tensors are handed over as raw device pointers (data_ptr())."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class CfsHipError(RuntimeError):
    code = 0


ERR_ARG, ERR_DEVICE, ERR_UNSUPPORTED, ERR_NOMEM, ERR_INTERNAL, ERR_MIRROR = -1, -2, -3, -4, -5, -6


class Options(C.Structure):
    _fields_ = [("max_slots", C.c_int), ("max_tile_nnz", C.c_int),
                ("block_threads", C.c_int), ("flags", C.c_int)]


class SymStats(C.Structure):
    _fields_ = [("n", C.c_int), ("row_begin", C.c_int), ("row_end", C.c_int),
                ("value_bytes", C.c_int), ("nnz_low", C.c_int64), ("nnz_diag", C.c_int64),
                ("nnz_full", C.c_int64), ("ntiles", C.c_int), ("nslices", C.c_int),
                ("max_slots_used", C.c_int), ("block_threads", C.c_int),
                ("halo_slots", C.c_int64), ("fold_rows", C.c_int64),
                ("remote_vals", C.c_int64), ("lds_bytes", C.c_int64),
                ("bytes_algorithmic", C.c_int64), ("bytes_streamed", C.c_int64),
                ("device_bytes", C.c_int64), ("mirror_entries", C.c_int64),
                ("far_entries", C.c_int64), ("ngroups", C.c_int), ("reserved_", C.c_int)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PlanReport(C.Structure):
    _fields_ = [("ntiles", C.c_int), ("ngroups", C.c_int), ("lds_slots", C.c_int),
                ("nslices", C.c_int64), ("halo_slots", C.c_int64), ("stream_len", C.c_int64),
                ("nnz_low", C.c_int64), ("fold_rows", C.c_int64), ("remote_vals", C.c_int64),
                ("decoded", C.c_int64), ("mismatches", C.c_int64),
                ("mirror_entries", C.c_int64), ("far_entries", C.c_int64)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/cfs_hip.h declares (tests check the .so exports them all)
SYMBOLS = [
    "cfs_hip_abi_version", "cfs_hip_last_error", "cfs_hip_device_count", "cfs_hip_init",
    "cfs_hip_current_device", "cfs_hip_runtime_bound", "cfs_hip_pinned_owns", "cfs_hip_pinned_pool_stats", "cfs_hip_default_stream", "cfs_hip_synchronize", "cfs_hip_alloc", "cfs_hip_free",
    "cfs_hip_memcpy", "cfs_hip_memset", "cfs_hip_sym_create_f64", "cfs_hip_sym_create_f32",
    "cfs_hip_sym_create_shard_f64", "cfs_hip_sym_create_shard_f32",
    "cfs_hip_sym_create_multi_f64", "cfs_hip_sym_create_multi_f32", "cfs_hip_comm_create", "cfs_hip_comm_info", "cfs_hip_comm_destroy", "cfs_hip_comm_reduce_scatter",
    "cfs_hip_comm_allgather", "cfs_hip_comm_wait_consumed", "cfs_hip_sym_num_gpus", "cfs_hip_sym_multi_set_xmode", "cfs_hip_sym_multi_devices", "cfs_hip_sym_balanced_splits", "cfs_hip_sym_destroy", "cfs_hip_sym_update_values_f64", "cfs_hip_sym_update_values_f32", "cfs_hip_sym_spmv",
    "cfs_hip_sym_spmv_async", "cfs_hip_sym_cg", "cfs_hip_sym_shard_send_counts", "cfs_hip_sym_shard_send_rows",
    "cfs_hip_sym_shard_set_recv", "cfs_hip_sym_spmv_local_async",
    "cfs_hip_sym_recv_fold_async", "cfs_hip_sym_spmv_phases_async", "cfs_hip_sym_get_stats", "cfs_hip_sym_debug_digest", "cfs_hip_sym_debug_plan_note", "cfs_hip_sym_debug_timeline", "cfs_hip_sym_debug_group_features", "cfs_hip_sym_plan_check_f64",
    "cfs_hip_sym_plan_check_f32", "cfs_hip_sym_plan_send_info_f64", "cfs_hip_csr_create_f64", "cfs_hip_csr_create_f32",
    "cfs_hip_csr_spmv", "cfs_hip_csr_spmv_async", "cfs_hip_csr_destroy", "cfs_hip_csr_kernel_form", "cfs_hip_csr_stats",
    "cfs_hip_event_create", "cfs_hip_event_record", "cfs_hip_event_elapsed_ms",
    "cfs_hip_event_destroy",
]


def lib_path():
    # CFS_HIP_LIB: developer override to A/B an experimental build of the library
    return os.environ.get("CFS_HIP_LIB") or os.path.join(HERE, "libcfs_hip.so")


def load():
    """Load libcfs_hip.so.  There is NO fallback: if the HIP extension is
    missing the product path fails here, loudly."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise CfsHipError(
            f"{path} is missing: the HIP extension is not built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback for the product path.")
    lib = C.CDLL(path)
    vp, ip, i64 = C.c_void_p, C.POINTER(C.c_int), C.c_int64
    lib.cfs_hip_last_error.restype = C.c_char_p
    lib.cfs_hip_alloc.argtypes = [C.c_size_t, C.c_int, C.POINTER(vp)]
    lib.cfs_hip_free.argtypes = [vp, C.c_int]
    lib.cfs_hip_current_device.argtypes = [ip]
    lib.cfs_hip_pinned_owns.argtypes = [vp]
    lib.cfs_hip_pinned_pool_stats.argtypes = [C.POINTER(C.c_size_t)] * 3
    lib.cfs_hip_memcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
    lib.cfs_hip_memset.argtypes = [vp, C.c_int, C.c_size_t]
    lib.cfs_hip_default_stream.argtypes = [C.POINTER(vp)]
    lib.cfs_hip_synchronize.argtypes = [vp]
    for suf in ("f64", "f32"):
        getattr(lib, "cfs_hip_sym_create_" + suf).argtypes = [
            C.c_int, vp, vp, vp, C.POINTER(Options), C.POINTER(vp)]
        getattr(lib, "cfs_hip_sym_create_shard_" + suf).argtypes = [
            C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, C.POINTER(Options), C.POINTER(vp)]
        getattr(lib, "cfs_hip_sym_create_multi_" + suf).argtypes = [
            C.c_int, vp, vp, vp, C.c_int, vp, C.POINTER(Options), C.POINTER(vp)]
        getattr(lib, "cfs_hip_sym_plan_check_" + suf).argtypes = [
            C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, C.POINTER(Options),
            C.POINTER(PlanReport)]
        getattr(lib, "cfs_hip_csr_create_" + suf).argtypes = [
            C.c_int, C.c_int, vp, vp, vp, C.POINTER(vp)]
    lib.cfs_hip_sym_plan_send_info_f64.argtypes = [
        C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, C.POINTER(Options), vp, vp, C.c_int, ip]
    lib.cfs_hip_sym_balanced_splits.argtypes = [C.c_int, vp, vp, C.c_int, vp]
    lib.cfs_hip_sym_num_gpus.argtypes = [vp, ip]
    if hasattr(lib, "cfs_hip_comm_create"):
        lib.cfs_hip_comm_create.argtypes = [C.c_int, vp, C.c_int, C.POINTER(vp)]
        lib.cfs_hip_comm_info.argtypes = [vp, ip, ip]
        lib.cfs_hip_comm_destroy.argtypes = [vp]
        lib.cfs_hip_comm_reduce_scatter.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, vp]
        lib.cfs_hip_comm_allgather.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, vp]
        lib.cfs_hip_comm_wait_consumed.argtypes = [vp, C.c_int, vp]
    if hasattr(lib, "cfs_hip_sym_multi_set_xmode"):
        lib.cfs_hip_sym_multi_set_xmode.argtypes = [vp, C.c_int]
        lib.cfs_hip_sym_multi_devices.argtypes = [vp, vp, C.c_int, ip]
    lib.cfs_hip_sym_destroy.argtypes = [vp]
    lib.cfs_hip_sym_update_values_f64.argtypes = [vp, vp, C.c_longlong]
    lib.cfs_hip_sym_update_values_f32.argtypes = [vp, vp, C.c_longlong]
    lib.cfs_hip_sym_spmv.argtypes = [vp, vp, vp]
    lib.cfs_hip_sym_spmv_async.argtypes = [vp, vp, vp, vp]
    lib.cfs_hip_sym_shard_send_counts.argtypes = [vp, vp]
    lib.cfs_hip_sym_shard_send_rows.argtypes = [vp, vp]
    lib.cfs_hip_sym_shard_set_recv.argtypes = [vp, C.c_int, vp]
    lib.cfs_hip_sym_spmv_local_async.argtypes = [vp, vp, vp, vp, vp]
    lib.cfs_hip_sym_recv_fold_async.argtypes = [vp, vp, vp, vp]
    lib.cfs_hip_sym_spmv_phases_async.argtypes = [vp, vp, vp, vp, C.c_int, vp]
    lib.cfs_hip_sym_get_stats.argtypes = [vp, C.POINTER(SymStats)]
    lib.cfs_hip_sym_debug_timeline.argtypes = [vp, vp, vp, vp, C.c_int, ip]
    if hasattr(lib, "cfs_hip_sym_debug_digest"):  # (absent from older builds loaded through CFS_HIP_LIB)
        lib.cfs_hip_sym_debug_digest.argtypes = [vp, vp, C.c_int]
        lib.cfs_hip_sym_debug_plan_note.argtypes = [vp, C.c_char_p, C.c_int]
    lib.cfs_hip_sym_debug_group_features.argtypes = [vp, vp, C.c_int, ip]
    lib.cfs_hip_csr_spmv.argtypes = [vp, vp, vp]
    lib.cfs_hip_csr_spmv_async.argtypes = [vp, vp, vp, vp]
    lib.cfs_hip_csr_destroy.argtypes = [vp]
    if hasattr(lib, "cfs_hip_csr_kernel_form"):
        lib.cfs_hip_csr_kernel_form.argtypes = [vp, ip, ip]
    if hasattr(lib, "cfs_hip_sym_cg"):
        lib.cfs_hip_sym_cg.argtypes = [vp, vp, vp, C.c_double, C.c_int, C.c_int, ip, C.POINTER(C.c_double), vp]
    if hasattr(lib, "cfs_hip_csr_stats"):
        lib.cfs_hip_csr_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.cfs_hip_event_create.argtypes = [C.POINTER(vp)]
    lib.cfs_hip_event_record.argtypes = [vp, vp]
    lib.cfs_hip_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
    lib.cfs_hip_event_destroy.argtypes = [vp]
    _LIB = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().cfs_hip_last_error().decode(errors="replace")
        e = CfsHipError(f"cfs_hip error {rc}: {msg}")
        e.code = rc
        raise e
