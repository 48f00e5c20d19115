"""cfs_spmv_amd -- MI355X-native symmetric SpMV (the hot path of athelaf/cfs-spmv).

(The task names the package `cfs-spmv_amd`; a hyphen is not importable, so the
directory is `cfs_spmv_amd`.)

  csrc/cfs_hip.hip, csrc/cfs_plan.hpp  kernels, tile schedule, C ABI -> libcfs_hip.so
  csrc/cfs_devplan.hpp                 tune() on the GPU: the schedule built by HIP kernels
  _lib.py                              ctypes binding of include/cfs_hip.h
  matrix.py                            SparseMatrix / SpDMV mirror over the C ABI
  dist.py                              1-D row-block sharding over torch.distributed
  solver.py                            solver-style caller (CG): every product fed back as the next x
  synth.py, csrc/cfs_synth.c           synthetic SuiteSparse stand-ins (workload only)
"""
from ._lib import CfsHipError, load, lib_path  # noqa: F401
from .matrix import (FLAG_HOST_PLAN, FLAG_KEEP_VALUE_MAP, FLAG_SHARD_EXCHANGE, CsrMatrix, Format, Kernel, SpDMV, SymMatrix, Tuning,  # noqa: F401
                     balanced_splits, make_options, plan_check, plan_send_info)
