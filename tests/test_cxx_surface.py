"""The C++ host surface (include/cfs.hpp, libsparse.so, the two drivers).

CPU part: the parallel Matrix-Market reader of libsparse yields the same CSR as
the oracle's restatement of the reference reader (which is itself pinned to the
genuine reader, test_oracle_mmf.py) on every fixture and on a structured
stand-in.  GPU part: the drivers run end to end with the reference's CLI."""
import ctypes as C
import glob
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
FILES = sorted(glob.glob(os.path.join(GOLD, "*.mtx")))


def cxx_load(path):
    lib = C.CDLL(os.path.join(ROOT, "build", "libsparse.so"))
    n, m, sym, nnz = C.c_int(), C.c_int(), C.c_int(), C.c_long()
    rp, ci, va = C.c_void_p(), C.c_void_p(), C.c_void_p()
    err = C.create_string_buffer(256)
    rc = lib.cfs_mmf_load_csr_f64(os.fsencode(path), C.byref(n), C.byref(m), C.byref(nnz),
                                  C.byref(sym), C.byref(rp), C.byref(ci), C.byref(va), err, 256)
    if rc != 0:
        raise ValueError(err.value.decode())
    lib.cfs_mmf_free.argtypes = [C.c_void_p]

    def take(p, cnt, dt):
        nb = cnt * np.dtype(dt).itemsize
        a = np.frombuffer((C.c_char * nb).from_address(p.value), dtype=dt, count=cnt).copy() \
            if cnt else np.zeros(0, dt)
        lib.cfs_mmf_free(p)
        return a
    return dict(nrows=n.value, ncols=m.value, nnz=nnz.value, symmetric=bool(sym.value),
                rowptr=take(rp, n.value + 1, np.int32), colind=take(ci, nnz.value, np.int32),
                values=take(va, nnz.value, np.float64))


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_parallel_reader_equals_reference_reader(path):
    a, b = cxx_load(path), oracle.mmf_load(path)
    for k in ("nrows", "ncols", "nnz", "symmetric"):
        assert a[k] == b[k]
    assert np.array_equal(a["rowptr"], b["rowptr"])   # bit-exact row_ptr
    assert np.array_equal(a["colind"], b["colind"])   # bit-exact col_idx
    assert np.array_equal(a["values"].view(np.uint64), b["values"].view(np.uint64))


def test_parallel_reader_on_stand_in(tmp_path):
    from cfs_spmv_amd import synth
    n, rp, ci, va, _ = synth.generate("ldoor", 0.01)
    p = str(tmp_path / "m.mtx")
    synth.write_mtx(p, n, rp, ci, va)
    a = cxx_load(p)
    assert a["symmetric"] and a["nrows"] == n and a["nnz"] == rp[-1]
    assert np.array_equal(a["rowptr"], rp) and np.array_equal(a["colind"], ci)
    assert np.array_equal(a["values"], va)


def test_parallel_reader_leniency_and_errors(tmp_path):
    p = tmp_path / "t.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n2 2 2\n1\t1\t1.5\n2 2 2.5")
    a = cxx_load(str(p))  # tabs and a missing last newline are accepted here
    assert list(a["values"]) == [1.5, 2.5]
    p.write_text("%%MatrixMarket matrix coordinate real hermitian\n2 2 1\n1 1 1\n")
    with pytest.raises(ValueError, match="unsupported symmetry"):
        cxx_load(str(p))
    p.write_text("%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1\n")
    with pytest.raises(ValueError, match="mmf ended"):
        cxx_load(str(p))


def test_headers_keep_the_reference_surface():
    """names a caller of the reference uses (bench/bench_spmv_mmf.cpp:16-23,:100-164)"""
    inc = os.path.join(ROOT, "include")
    text = "".join(open(f).read() for f in glob.glob(os.path.join(inc, "**", "*.hpp"),
                                                    recursive=True))
    for needle in ["namespace cfs", "class SparseMatrix", "class CSRMatrix", "struct SpDMV",
                   "internal_alloc", "internal_free", "get_num_threads", "isEqual",
                   "enum class Platform", "enum class Format { none, csr, sss, hyb }",
                   "enum class Tuning { None, Aggressive }", "dense_vector_multiply",
                   "rowptr()", "colind()", "values()"]:
        assert needle in text, needle


@pytest.mark.gpu
def test_drivers_end_to_end(tmp_path):
    from cfs_spmv_amd import synth
    n, rp, ci, va, _ = synth.generate("pwtk", 0.05)
    p = str(tmp_path / "pwtk_like.mtx")
    synth.write_mtx(p, n, rp, ci, va)
    env = dict(os.environ, CFS_SEED="7")
    for fmt in ("0", "1", "2"):
        r = subprocess.run([os.path.join(ROOT, "build", "test_spmv_mmf"), p, fmt],
                           capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0 and "PASSED!" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([os.path.join(ROOT, "build", "bench_spmv_mmf"), p, "1", "64"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"matrix: pwtk_like.mtx format: SSS preproc\(sec\): (\S+) t\(sec\): (\S+) "
                  r"gflops/s: (\S+) threads: (\d+) size\(MB\): (\S+)", r.stdout)
    assert m, r.stdout
    assert float(m.group(3)) > 0
    # the single-precision build of the same driver (the reference's default configure)
    r = subprocess.run([os.path.join(ROOT, "build_sp", "bench_spmv_mmf"), p, "1", "64"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "format: SSS" in r.stdout, r.stdout + r.stderr
    # general file through Format::sss: silent fall-back to CSR (csr_matrix.tpp:13-19)
    g = str(tmp_path / "gen.mtx")
    synth.write_mtx(g, n, rp, ci, va, general=True)
    r = subprocess.run([os.path.join(ROOT, "build", "test_spmv_mmf"), g, "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "PASSED!" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_unschedulable_symmetric_matrix_falls_back_to_gpu_csr(tmp_path):
    """a dense diagonal block wider than the LDS window cannot be tiled: tune() says so
    and binds the general CSR kernel (still on the GPU); the self-check still passes"""
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    n, k = 3000, 300
    B = sp.random(n, n, density=0.002, random_state=3, format="lil")
    B[1000:1000 + k, 1000:1000 + k] = rng.uniform(-1, 0, (k, k))
    A = sp.tril(B.tocsr(), -1)
    A = (A + A.T + sp.diags(np.full(n, 400.0))).tocsr()
    A.sort_indices()
    p = str(tmp_path / "block.mtx")
    from cfs_spmv_amd import synth
    synth.write_mtx(p, n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data)
    env = dict(os.environ, CFS_SEED="7", CFS_HIP_MAX_SLOTS="128")
    r = subprocess.run([os.path.join(ROOT, "build", "test_spmv_mmf"), p, "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "PASSED!" in r.stdout, r.stdout + r.stderr
    assert "falling back to the general CSR kernel" in r.stdout, r.stdout


def test_binary_cache_roundtrip(tmp_path, monkeypatch):
    """CFS_MTX_CACHE_DIR: second load comes from <dir>/<file>.f64.csrbin, a touched
    source file invalidates it"""
    from cfs_spmv_amd import synth
    n, rp, ci, va, _ = synth.generate("pwtk", 0.02)
    p = str(tmp_path / "m.mtx")
    synth.write_mtx(p, n, rp, ci, va)
    cdir = tmp_path / "cache"
    cdir.mkdir()
    monkeypatch.setenv("CFS_MTX_CACHE_DIR", str(cdir))
    a = cxx_load(p)
    cache = cdir / "m.mtx.f64.csrbin"
    assert cache.exists()
    b = cxx_load(p)  # served by the cache
    for k in ("rowptr", "colind", "values"):
        assert np.array_equal(a[k], b[k])
    assert a["nnz"] == b["nnz"] and a["symmetric"] == b["symmetric"]
    # corrupt the cached values: proves the second path really reads the cache
    raw = bytearray(cache.read_bytes())
    raw[-8:] = np.float64(123.5).tobytes()
    cache.write_bytes(bytes(raw))
    c = cxx_load(p)
    assert c["values"][-1] == 123.5
    # a changed source file (size/mtime) invalidates the cache
    with open(p, "a") as f:
        f.write("% trailing comment\n")
    d = cxx_load(p)
    assert d["values"][-1] == a["values"][-1]


@pytest.mark.gpu
def test_bench_takes_a_real_file_from_cfs_mtx_dir(tmp_path):
    """the stand-ins are only used when $CFS_MTX_DIR/<name>.mtx is absent"""
    import json
    from cfs_spmv_amd import synth
    n, rp, ci, va, low = synth.generate("pwtk", 0.05)
    synth.write_mtx(str(tmp_path / "pwtk.mtx"), n, rp, ci, va)
    env = dict(os.environ, CFS_MTX_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--matrix", "pwtk",
                        "--steps", "10", "--warmup", "3", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["data"] == "file" and "pwtk.mtx" in d["config"]["workload"]
    assert f"n={n}," in d["config"]["workload"] and f"nnz_low={low}," in d["config"]["workload"]


def test_value_parser_is_bit_exact_with_strtod(tmp_path):
    """the reader's fast path for short decimals (Clinger) and its strtod fall-back give
    the correctly rounded double for every spelling -- what the reference's atof returns"""
    import random
    import struct
    rnd = random.Random(7)
    strs = ["0", "-0", "0.0", "1", "-1", "1e22", "1e23", "1e-22", "1e-23", "123456789012345",
            "1234567890123456", "0.000001", "5e-324", "1.7976931348623157e308", "+3.5", "1.",
            "-.5", "00012.5000", "9007199254740993", "0.1e1", "1E+2", "4.9406564584124654e-324",
            "2.2250738585072014e-308", "1e400", "-1e-400", "0.3", "0.1", "123456.789e-3"]
    fmts = ["%.17g", "%.15g", "%.6e", "%.3f", "%g", "%.12E", "%.16g", "%.1f"]
    for _ in range(20000):
        v = rnd.choice([rnd.uniform(-1, 1), rnd.uniform(-1e6, 1e6),
                        rnd.gauss(0, 1) * 10 ** rnd.randint(-30, 30),
                        float(rnd.randint(-10 ** 9, 10 ** 9))])
        strs.append(rnd.choice(fmts) % v)
    p = tmp_path / "values.mtx"
    with open(p, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n%d 1 %d\n" % (len(strs), len(strs)))
        for i, sv in enumerate(strs):
            f.write("%d 1 %s\n" % (i + 1, sv))
    got = cxx_load(str(p))["values"]
    exp = np.array([float(sv) for sv in strs])  # Python's float() is correctly rounded
    bad = [(strs[i], exp[i], got[i]) for i in range(len(strs))
           if struct.pack("d", exp[i]) != struct.pack("d", got[i])]
    assert not bad, bad[:5]
