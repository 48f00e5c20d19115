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


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_parallel_reader_is_clean_under_asan_ubsan(tmp_path):
    """golden fixtures plus malformed files: truncated lines, missing trailing newline,
    indices out of range, a size line that promises more than the body holds, junk"""
    exe = str(tmp_path / "asan_reader")
    srcs = [os.path.join(ROOT, "src", "mmf.cpp"), os.path.join(ROOT, "tests", "native", "asan_reader.cpp")]
    # get_host_threads() is the only thing the reader needs from runtime.cpp; give the
    # harness its own so that it links without the HIP library
    shim = tmp_path / "threads.cpp"
    shim.write_text("namespace cfs { namespace util { namespace runtime { int get_host_threads() { return 4; } } } }\n")
    cmd = ["g++", "-std=c++11", "-O1", "-g", "-fopenmp", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "include")] + srcs + \
          [str(shim), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    gold = os.path.join(ROOT, "tests", "golden")
    files = sorted(os.path.join(gold, f) for f in os.listdir(gold) if f.endswith(".mtx"))
    bad = {
        "trunc.mtx": "%%MatrixMarket matrix coordinate real symmetric\n5 5 4\n1 1 1.0\n2 1\n3",
        "nonl.mtx": "%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 1.0\n3 2 2.5",
        "range.mtx": "%%MatrixMarket matrix coordinate real general\n3 3 2\n7 1 1.0\n0 9 2.0\n",
        "short.mtx": "%%MatrixMarket matrix coordinate real symmetric\n4 4 100\n2 1 1.0\n",
        "junk.mtx": "hello world\n\n\n1 2 three\n" + "x" * 5000,
        "empty.mtx": "",
        "sizeonly.mtx": "%%MatrixMarket matrix coordinate real general\n0 0 0\n",
        "neg.mtx": "%%MatrixMarket matrix coordinate real general\n-3 3 2\n1 1 1.0\n",
        "huge.mtx": "%%MatrixMarket matrix coordinate real general\n2147483647 2147483647 3\n1 1 1.0\n",
    }
    # a file whose size is an exact multiple of the page size and whose last token runs
    # to the end of the mapping (no trailing newline): an integer token (pattern entry)
    # and a 20-digit value that takes the strtod path
    def page_file(last_line):
        head = "%%MatrixMarket matrix coordinate real general\n"
        body = "9 9 3\n1 1 1.0\n2 2 2.0\n" + last_line
        pad = 4096 - (len(head) + len(body)) - 2
        return head + "%" + "p" * pad + "\n" + body
    bad["page_int.mtx"] = page_file("9 8")
    bad["page_dbl.mtx"] = page_file("9 8 1.2345678901234567890")
    assert len(bad["page_int.mtx"]) == 4096 and len(bad["page_dbl.mtx"]) == 4096
    for name, text in bad.items():
        (tmp_path / name).write_text(text)
        files.append(str(tmp_path / name))
    env = dict(os.environ, OMP_NUM_THREADS="4", ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([exe] + files, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout + r.stderr)[-4000:]
