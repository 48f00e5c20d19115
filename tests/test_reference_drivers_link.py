"""Drop-in guard: the reference's OWN, unmodified drivers compile and link against
this repo's headers and libsparse.so.

Build-container only: the driver sources are compiled WHERE THEY LIE under
/root/reference (nothing is copied into the repo, the binaries go to a temporary
directory and are never run here -- no GPU -- nor shipped to the GPU box, where
/root/reference does not exist and the test skips).

What it pins: `#include "cfs.hpp"` gives the drivers everything
the reference's header set gives them -- names, namespaces, signatures and the
transitive standard headers (`<random>` through matrix/sparse_matrix.hpp:4,
used at bench/bench_spmv_mmf.cpp:123-125 and test/test_spmv_mmf.cpp:73-75)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
DRIVERS = ["bench/bench_spmv_mmf.cpp", "test/test_spmv_mmf.cpp"]

pytestmark = pytest.mark.skipif(
    not all(os.path.exists(os.path.join(REF, d)) for d in DRIVERS) or shutil.which("g++") is None,
    reason="needs /root/reference (build container only) and g++")


@pytest.mark.parametrize("precision", ["single", "double"])
@pytest.mark.parametrize("driver", DRIVERS)
def test_unmodified_reference_driver_compiles_and_links(driver, precision, tmp_path):
    # the reference's own flags: -std=c++11 -fopenmp (configure.ac:19,27), --enable-dp -> -D_USE_DOUBLE
    build = "build" if precision == "double" else "build_sp"
    lib = os.path.join(ROOT, build, "libsparse.so")
    assert os.path.exists(lib), f"{lib} missing: run __graft_entry__.build()"
    out = str(tmp_path / "drv")
    cmd = ["g++", "-std=c++11", "-fopenmp", "-Wall"]
    if precision == "double":
        cmd.append("-D_USE_DOUBLE")
    cmd += ["-I" + os.path.join(ROOT, "include"), os.path.join(REF, driver), "-o", out,
            "-L" + os.path.join(ROOT, build), "-lsparse",
            "-L" + os.path.join(ROOT, "cfs_spmv_amd"), "-lcfs_hip",
            "-Wl,-rpath," + os.path.join(ROOT, build),
            "-Wl,-rpath," + os.path.join(ROOT, "cfs_spmv_amd")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert os.path.getsize(out) > 0
    # every undefined cfs:: symbol of the driver is resolved by libsparse.so (a link
    # with --no-undefined semantics: executables already fail on unresolved symbols)
    nm = subprocess.run(["nm", "-C", "--undefined-only", out], capture_output=True, text=True).stdout
    wanted = [l for l in nm.splitlines() if "cfs::" in l]
    assert wanted, "the driver should import cfs:: symbols from libsparse.so"


def test_reference_header_set_is_mirrored():
    """every header the reference installs (src/Makefile.am:3) exists under include/"""
    ref_inc = os.path.join(REF, "include")
    missing = []
    for dp, _, fs in os.walk(ref_inc):
        for f in fs:
            if f.endswith(".hpp"):
                rel = os.path.relpath(os.path.join(dp, f), ref_inc)
                if not os.path.exists(os.path.join(ROOT, "include", rel)):
                    missing.append(rel)
    # .tpp bodies are implementation (ours live in src/*.cpp); headers must all be there
    assert not missing, missing
