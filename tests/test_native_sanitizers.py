"""The host-side schedule builder under AddressSanitizer + UBSan (CPU build only: GPU
sanitizers are not available on the pool).  tests/native/asan_plan.cpp builds and
decodes schedules for node-block, banded and hub matrices in every order / shard form."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_schedule_builder_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "asan_plan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fopenmp", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "cfs_spmv_amd", "csrc"),
           os.path.join(ROOT, "tests", "native", "asan_plan.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, OMP_NUM_THREADS="4", ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout + r.stderr)[-4000:]
