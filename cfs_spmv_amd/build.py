"""In-tree build of the native pieces (no JIT cache: the .so files travel with
the repo snapshot to the GPU box).

  libcfs_hip.so   hipcc --offload-arch=gfx950   kernels + C ABI (include/cfs_hip.h)
  libcfs_synth.so gcc                           synthetic workload generator
  oracle/         make                          CPU oracle (test infrastructure)
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, cwd=None):
    print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=cwd, check=True)


def hipcc_path():
    for p in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if p and os.path.exists(p):
            return p
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def build_hip(force=False):
    out = os.path.join(HERE, "libcfs_hip.so")
    srcs = [os.path.join(CSRC, "cfs_hip.hip"), os.path.join(CSRC, "cfs_plan.hpp"),
            os.path.join(CSRC, "cfs_devplan.hpp"), os.path.join(CSRC, "cfs_comm.hpp"), os.path.join(CSRC, "cfs_csr.hpp"), os.path.join(CSRC, "cfs_solver.hpp"),
            os.path.join(CSRC, "cfs_runtime.hpp"), os.path.join(ROOT, "include", "cfs_hip.h")]
    if force or _newer(out, srcs):
        _run([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
              "-shared", "-fopenmp", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
              srcs[0], "-o", out, "-ldl"])
    return out


def build_synth(force=False):
    out = os.path.join(HERE, "libcfs_synth.so")
    src = os.path.join(CSRC, "cfs_synth.c")
    if force or _newer(out, [src]):
        _run(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-std=gnu11", src, "-o", out, "-lm"])
    return out


def build_oracle(force=False):
    odir = os.path.join(ROOT, "oracle")
    if force:
        _run(["make", "-C", odir, "clean"])
    _run(["make", "-C", odir])
    return os.path.join(odir, "liboracle.so")


def build_cxx(force=False):
    """libsparse.so + bench_spmv_mmf + test_spmv_mmf (the reference's products)"""
    if force:
        _run(["make", "-C", ROOT, "clean"])
        _run(["make", "-C", ROOT, "clean", "BUILD=build_sp"])
    _run(["make", "-C", ROOT, "DP=1"])
    # the reference's default configure is single precision (the Queen_4147 config)
    _run(["make", "-C", ROOT, "BUILD=build_sp"])
    return os.path.join(ROOT, "build", "libsparse.so")


def build_all(force=False):
    build_hip(force)
    build_synth(force)
    build_oracle(force)
    build_cxx(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
