"""GPU tests of the round-2 features, all through the C ABI:
Format::hyb (far entries), the deterministic mode, the multi-device handle behind
CFS_NUM_GPUS, the pinned-host pool of the host-pointer path, per-device runtime state,
the non-regular stand-in."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import scaled_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FLAG_NO_REORDER, FLAG_CLUSTER, FLAG_NO_CAL, FLAG_HYB, FLAG_NO_HYB, FLAG_DET = 8, 16, 32, 128, 256, 1024
TOL = {np.float64: 1e-12, np.float32: 1e-5}


def _torch():
    import torch
    assert torch.cuda.is_available(), "gpu test without a GPU"
    return torch


def _spmv(A, x, torch, garbage=777.0):
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    yd = torch.full((A.nrows(),), garbage, dtype=xd.dtype, device="cuda")
    A.dense_vector_multiply(yd, xd)
    torch.cuda.synchronize()
    return yd.cpu().numpy()


def _arrow(n=4000, hubs=3, seed=2):
    """banded matrix + a few hub rows that touch a column of every 7th row"""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for i in range(1, n):
        c = np.unique(rng.integers(max(0, i - 60), i, size=min(i, 9)))
        rows.append(np.full(c.size, i))
        cols.append(c)
    for h in range(hubs):
        i = n - 1 - 17 * h
        c = np.arange(h, i, 7)
        rows.append(np.full(c.size, i))
        cols.append(c)
    r, c = np.concatenate(rows), np.concatenate(cols)
    L = sp.coo_matrix((rng.uniform(-1, 1, r.size), (r, c)), shape=(n, n)).tocsr()  # sums duplicates
    A = (L + L.T + sp.diags(rng.uniform(1, 2, n))).tocsr()
    A.sort_indices()
    return n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data


def _scatter(n=30000, seed=5):
    """every row: a short band + 6 single entries scattered over the whole matrix (the
    single-use halo columns HYB is for)"""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for i in range(1, n):
        c = np.unique(np.concatenate([rng.integers(max(0, i - 12), i, size=min(i, 5)),
                                      rng.integers(0, i, size=min(i, 6))]))
        rows.append(np.full(c.size, i))
        cols.append(c)
    r, c = np.concatenate(rows), np.concatenate(cols)
    L = sp.coo_matrix((rng.uniform(-1, 1, r.size), (r, c)), shape=(n, n)).tocsr()
    A = (L + L.T + sp.diags(rng.uniform(1, 2, n))).tocsr()
    A.sort_indices()
    return n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", ["ldoor", "arrow", "scatter", "pdb1HYS", "unstruct"])
@pytest.mark.parametrize("flags", [FLAG_HYB, FLAG_HYB | FLAG_NO_REORDER, FLAG_HYB | FLAG_CLUSTER])
def test_hyb_parity(shape, flags, dtype):
    """Format::hyb: far entries (stored by both tiles they touch, processed one-sided
    with x gathered from global memory) against the oracle, both row orders"""
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import synth
    from oracle import oracle
    torch = _torch()
    if shape == "arrow":
        n, rp, ci, va = _arrow()
    elif shape == "scatter":
        n, rp, ci, va = _scatter()
    else:
        n, rp, ci, va, _ = synth.generate(shape, {"ldoor": 0.15, "pdb1HYS": 0.5, "unstruct": 0.03}[shape])
    va = va.astype(dtype)
    x = np.random.default_rng(9).uniform(-1, 1, n).astype(dtype)
    A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=flags | FLAG_NO_CAL))
    st = A.stats()
    y = _spmv(A, x, torch)
    y2 = _spmv(A, x, torch, garbage=-1.5)
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    assert scaled_err(y, y_ld, absrow) <= TOL[dtype]
    assert scaled_err(y2, y_ld, absrow) <= TOL[dtype]
    o = oracle.SymOracle(n, rp, ci, va, 3)
    assert scaled_err(y, o.spmv(x).astype(np.float64), absrow) <= TOL[dtype]
    o.close()
    if shape in ("ldoor", "scatter", "pdb1HYS"):
        assert st["far_entries"] > 0  # the path really ran
        B = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=(flags & ~FLAG_HYB) | FLAG_NO_CAL))
        assert st["halo_slots"] < B.stats()["halo_slots"]
        B.close()
    A.close()


def test_tune_chooses_hyb_by_measurement_and_no_hyb_forbids_it():
    import cfs_spmv_amd as cfs
    n, rp, ci, va = _scatter(n=400000, seed=6)  # > 2M stored nonzeros: tune() measures
    A = cfs.SymMatrix(n, rp, ci, va)
    B = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=FLAG_NO_HYB))
    assert B.stats()["far_entries"] == 0
    torch = _torch()
    x = np.random.default_rng(1).uniform(-1, 1, n)
    ya, yb = _spmv(A, x, torch), _spmv(B, x, torch)
    from oracle import oracle
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    assert scaled_err(ya, y_ld, absrow) <= 1e-12 and scaled_err(yb, y_ld, absrow) <= 1e-12
    A.close()
    B.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name,scale,flags", [("pwtk", 0.2, 0), ("ldoor", 0.1, 0), ("Flan_1565", 0.05, 0),
                                              ("Flan_1565", 0.05, FLAG_NO_REORDER), ("pdb1HYS", 1.0, 0),
                                              ("ldoor", 0.1, FLAG_HYB), ("unstruct", 0.05, FLAG_HYB | FLAG_NO_REORDER)])
def test_deterministic_mode_is_bit_reproducible(name, scale, flags, dtype):
    """CFS_HIP_FLAG_DETERMINISTIC: run the SpMV several times (the run-twice protocol of
    test/test_spmv_mmf.cpp:82-83) and on a second handle of the same matrix: identical
    BITS, and the same tolerance against the oracle as the default mode.  (The default
    mode's LDS float atomics make the last bits of y vary from run to run.)"""
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import synth
    from oracle import oracle
    torch = _torch()
    n, rp, ci, va, _ = synth.generate(name, scale)
    va = va.astype(dtype)
    x = np.random.default_rng(3).uniform(-1, 1, n).astype(dtype)  # mixed signs
    opt = cfs.make_options(flags=flags | FLAG_DET | FLAG_NO_CAL)
    A = cfs.SymMatrix(n, rp, ci, va, options=opt)
    ys = [_spmv(A, x, torch, garbage=float(k)) for k in range(5)]
    for y in ys[1:]:
        assert np.array_equal(ys[0].view(np.uint8), y.view(np.uint8))
    A2 = cfs.SymMatrix(n, rp, ci, va, options=opt)
    assert np.array_equal(ys[0].view(np.uint8), _spmv(A2, x, torch).view(np.uint8))
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    assert scaled_err(ys[0], y_ld, absrow) <= TOL[dtype]
    assert A.stats()["block_threads"] == 512
    # far entries (Format::hyb) are part of the deterministic build too: their x values, gathered
    # from outside the window, enter the tile's scale
    assert (A.stats()["far_entries"] > 0) == bool(flags & FLAG_HYB)
    if flags & FLAG_HYB:
        # a few x values 2^40 above the others -- beyond the 23 bits of headroom of the fixed-point
        # words: a tile that gathers one of them as a far entry must have it in its scale, or the
        # sums overflow.  Same bits every run; a row whose own x values are 2^-40 of its tile's
        # largest keeps 2^-80 of THAT scale (the mode's contract, include/cfs_hip.h): 1e-9 of its own
        x2 = x.copy()
        x2[:: max(1, n // 97)] *= dtype(2.0 ** 40)
        yb = [_spmv(A, x2, torch, garbage=float(k)) for k in range(3)]
        assert np.array_equal(yb[0].view(np.uint8), yb[1].view(np.uint8))
        assert np.array_equal(yb[0].view(np.uint8), yb[2].view(np.uint8))
        y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x2)
        assert scaled_err(yb[0], y_ld, absrow) <= max(TOL[dtype], 1e-9)
    A.close()
    A2.close()


def test_deterministic_mode_with_wide_value_range():
    """the contract of the fixed-point sums: a contribution keeps 2^-68 of (max|a| of its
    tile) x (max|x| of the tile's window).  Rows scaled 2^-6 .. 2^6 against each other
    (entries spread over 2^24) still meet the 1e-12 tolerance; bit-reproducible"""
    import scipy.sparse as sp
    import cfs_spmv_amd as cfs
    from oracle import oracle
    torch = _torch()
    n = 5000
    rng = np.random.default_rng(4)
    L = sp.random(n, n, density=0.004, random_state=7, format="csr")
    L = sp.tril(L, -1).tocsr()
    scale = 2.0 ** rng.integers(-6, 7, size=n)
    D = sp.diags(scale)
    L = (D @ L @ D).tocsr()
    A = (L + L.T + sp.diags(scale * scale)).tocsr()
    A.sort_indices()
    rp, ci, va = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data
    x = rng.uniform(-1, 1, n)
    M = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=FLAG_DET))
    y = _spmv(M, x, torch)
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    assert scaled_err(y, y_ld, absrow) <= 1e-12
    assert np.array_equal(y.view(np.uint8), _spmv(M, x, torch).view(np.uint8))
    M.close()


@pytest.mark.parametrize("block", [0, 1024])
def test_deterministic_mode_is_independent_of_row_scaling(block):
    """per-SLOT fixed-point scales (row 1-norm x window max|x|): rows scaled 2^-30 .. 2^30 against
    each other -- entries spread over 2^120, far beyond what one scale per tile could hold -- meet
    the 1e-12 bound against the long-double row sums on their OWN scale, bit-reproducibly; also in
    the 1 024-thread shape.  Reference behaviour this stands in for: the fixed
    accumulation order of csr_matrix.tpp:3005-3013."""
    import scipy.sparse as sp
    import cfs_spmv_amd as cfs
    from oracle import oracle
    torch = _torch()
    n = 6000
    rng = np.random.default_rng(9)
    L = sp.tril(sp.random(n, n, density=0.004, random_state=11, format="csr"), -1).tocsr()
    scale = 2.0 ** rng.integers(-30, 31, size=n)
    D = sp.diags(scale)
    L = (D @ L @ D).tocsr()
    A = (L + L.T + sp.diags(scale * scale)).tocsr()
    A.sort_indices()
    rp, ci, va = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data
    x = rng.uniform(-1, 1, n)
    M = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(block_threads=block, flags=FLAG_DET | FLAG_NO_CAL))
    assert M.stats()["block_threads"] == (block or 512)
    y = _spmv(M, x, torch)
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    assert scaled_err(y, y_ld, absrow) <= 1e-12
    assert np.array_equal(y.view(np.uint8), _spmv(M, x, torch).view(np.uint8))
    M2 = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(block_threads=block, flags=FLAG_DET | FLAG_NO_CAL))
    assert np.array_equal(y.view(np.uint8), _spmv(M2, x, torch).view(np.uint8))
    M.close()
    M2.close()


@pytest.mark.parametrize("where", ["x", "a"])
@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_deterministic_mode_propagates_nan_and_inf(where, bad):
    """a NaN / Inf in x or among the values has no fixed-point image: the rows it reaches must
    read NaN / Inf (as the floating-point path gives), never plausible finite numbers -- a
    diverging solver on a deterministic handle has to see that it diverged"""
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import synth
    torch = _torch()
    n, rp, ci, va, _ = synth.generate("pwtk", 0.02)
    va = va.copy()
    x = synth.make_x(n)
    i = n // 2
    if where == "x":
        x[i] = bad
    else:
        lo = [j for j in range(rp[i], rp[i + 1]) if ci[j] < i]
        j = lo[len(lo) // 2]
        c = ci[j]
        va[j] = bad  # the stored lower entry (i, c) ...
        for q in range(rp[c], rp[c + 1]):  # ... and its image, so that the matrix stays symmetric
            if ci[q] == i:
                va[q] = bad
    y_plain = _spmv(cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=FLAG_NO_CAL)), x, torch)
    y_det = _spmv(cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=FLAG_DET | FLAG_NO_CAL)), x, torch)
    reached = ~np.isfinite(y_plain)  # the rows the bad number reaches in the default mode
    assert reached.any()
    assert not np.isfinite(y_det[reached]).any(), "finite garbage where the default mode has NaN / Inf"
    # ... and rows of tiles it does not reach are still numbers
    assert np.isfinite(y_det).sum() > 0.5 * n


@pytest.mark.parametrize("ngpus", [2, 3])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_multi_device_handle_on_one_device(ngpus, dtype):
    """cfs_hip_sym_create_multi_* (what CSRMatrix::tune builds for CFS_NUM_GPUS): N
    mirrored shards, here all on cuda:0, each on its own stream, driven by one thread"""
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import _lib, synth
    from oracle import oracle
    torch = _torch()
    n, rp, ci, va, _ = synth.generate("Flan_1565", 0.03)
    va = va.astype(dtype)
    x = synth.make_x(n, 42, dtype)
    A = cfs.SymMatrix(n, rp, ci, va, ngpus=ngpus)
    k = C.c_int()
    _lib.check(_lib.load().cfs_hip_sym_num_gpus(A._h, C.byref(k)))
    assert k.value == ngpus
    st = A.stats()
    assert st["row_begin"] == 0 and st["row_end"] == n and st["nnz_full"] == rp[-1]
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    for _ in range(3):
        assert scaled_err(_spmv(A, x, torch), y_ld, absrow) <= TOL[dtype]
    # host pointers through the same handle (staged, synchronous)
    yh = np.full(n, 3.0, dtype=dtype)
    A.dense_vector_multiply_host(yh, x)
    assert scaled_err(yh, y_ld, absrow) <= TOL[dtype]
    # how shards reach x / y (cfs_hip_sym_multi_set_xmode): replicated x + local y block copied
    # home (REPLICATE_ALL = 2 forces the copies for shards on the home device too: the copy
    # path of a multi-GPU node, exercised on this one device), and peer access (0); all three
    # compute the same product
    lib = _lib.load()
    nd = C.c_int()
    devs = (C.c_int * ngpus)()
    _lib.check(lib.cfs_hip_sym_multi_devices(A._h, devs, ngpus, C.byref(nd)))
    assert nd.value == 1 and list(devs) == [0] * ngpus
    for mode in (2, 0, 1):
        _lib.check(lib.cfs_hip_sym_multi_set_xmode(A._h, mode))
        for garbage in (1.0, -2.0):
            assert scaled_err(_spmv(A, x, torch, garbage=garbage), y_ld, absrow) <= TOL[dtype], mode
    A.close()


def test_cxx_driver_with_cfs_num_gpus(tmp_path):
    """the reference's self-check driver, unmodified command line, CFS_NUM_GPUS=2 (two
    shards on the one visible device), SSS and HYB; the bench driver prints gpus: 2"""
    from cfs_spmv_amd import synth
    n, rp, ci, va, _ = synth.generate("ldoor", 0.05)
    p = str(tmp_path / "ldoor_like.mtx")
    synth.write_mtx(p, n, rp, ci, va)
    env = dict(os.environ, CFS_SEED="11", CFS_NUM_GPUS="2")
    for fmt in ("1", "2"):
        r = subprocess.run([os.path.join(ROOT, "build", "test_spmv_mmf"), p, fmt],
                           capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0 and "PASSED!" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([os.path.join(ROOT, "build", "bench_spmv_mmf"), p, "2", "32"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "format: HYB" in r.stdout and "gpus: 2" in r.stdout, r.stdout + r.stderr
    assert "devices: 1" in r.stdout  # hbm_pct is taken against the DISTINCT devices (one here)
    # the copy path of a multi-GPU node (replicated x, y blocks copied home) behind the drivers
    env["CFS_MULTI_X"] = "replicate_all"
    r = subprocess.run([os.path.join(ROOT, "build", "test_spmv_mmf"), p, "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "PASSED!" in r.stdout, r.stdout + r.stderr


def test_tune_measures_the_general_kernel_where_the_tile_format_has_no_locality(tmp_path):
    """CSRMatrix::tune(Aggressive) behind the unmodified self-check driver: a power-law graph
    (hub columns: nearly every column of a tile is a halo slot or a far entry) is timed with the
    symmetric schedule AND the general CSR kernel, the faster one kept; a mesh-like matrix never
    takes that detour.  Either way the reference's own check passes."""
    from cfs_spmv_amd import synth
    env = dict(os.environ, CFS_SEED="5")
    for name, scale, expect in (("powerlaw", 0.3, True), ("tetmesh", 0.05, False)):
        n, rp, ci, va, _ = synth.generate(name, scale)
        p = str(tmp_path / f"{name}.mtx")
        synth.write_mtx(p, n, rp, ci, va)
        r = subprocess.run([os.path.join(ROOT, "build", "test_spmv_mmf"), p, "1"],
                           capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and "PASSED!" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        took_csr = "using the general CSR kernel" in r.stdout
        assert took_csr == expect, (name, r.stdout[-1500:])
        # ... and the choice can be switched off
        if expect:
            r = subprocess.run([os.path.join(ROOT, "build", "test_spmv_mmf"), p, "1"], capture_output=True,
                               text=True, env=dict(env, CFS_NO_FORMAT_CHOICE="1"), timeout=600)
            assert r.returncode == 0 and "PASSED!" in r.stdout and "general CSR kernel" not in r.stdout


def test_host_pointer_path_uses_the_pinned_pool():
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import _lib, synth
    from oracle import oracle
    _torch()
    lib = _lib.load()
    n, rp, ci, va, _ = synth.generate("pwtk", 0.2)
    x = synth.make_x(n)
    A = cfs.SymMatrix(n, rp, ci, va)
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)

    def pool():
        a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
        lib.cfs_hip_pinned_pool_stats(C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value
    live0 = pool()[0]
    y = np.full(n, 9.0)
    A.dense_vector_multiply_host(y, x)  # pageable numpy arrays: staged through pinned blocks
    assert scaled_err(y, y_ld, absrow) <= 1e-12
    assert pool()[0] == live0 + 2  # one block for x, one for y, kept by the handle
    A.dense_vector_multiply_host(y, x)
    assert pool()[0] == live0 + 2  # allocated once per handle
    # vectors that already live in pinned memory are DMA-ed in place
    px, py = C.c_void_p(), C.c_void_p()
    _lib.check(lib.cfs_hip_alloc(n * 8, 1, C.byref(px)))
    _lib.check(lib.cfs_hip_alloc(n * 8, 1, C.byref(py)))
    assert lib.cfs_hip_pinned_owns(px) == 1
    xp = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_double)), shape=(n,))
    yp = np.ctypeslib.as_array(C.cast(py, C.POINTER(C.c_double)), shape=(n,))
    xp[:] = x
    yp[:] = -4.0
    _lib.check(lib.cfs_hip_sym_spmv(A._h, py, px))
    assert scaled_err(np.array(yp), y_ld, absrow) <= 1e-12
    A.close()
    assert pool()[0] == live0 + 2  # the handle's blocks went back to the pool ...
    spare = pool()[1]
    _lib.check(lib.cfs_hip_free(px, 1))
    _lib.check(lib.cfs_hip_free(py, 1))
    assert pool()[0] == live0 and pool()[1] == spare + 2  # ... and so do these, for reuse
    q = C.c_void_p()
    _lib.check(lib.cfs_hip_alloc(n * 8, 1, C.byref(q)))
    assert pool()[1] == spare + 1  # a released block was handed out again
    _lib.check(lib.cfs_hip_free(q, 1))


def test_runtime_binds_the_callers_device_and_handles_remember_theirs():
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import _lib, synth
    torch = _torch()
    lib = _lib.load()
    d = C.c_int(-1)
    _lib.check(lib.cfs_hip_current_device(C.byref(d)))
    assert d.value == torch.cuda.current_device()
    _lib.check(lib.cfs_hip_init(torch.cuda.current_device()))  # idempotent, keeps handles alive
    n, rp, ci, va, _ = synth.generate("pwtk", 0.05)
    A = cfs.SymMatrix(n, rp, ci, va)
    _lib.check(lib.cfs_hip_init(torch.cuda.current_device()))
    x = synth.make_x(n)
    y = _spmv(A, x, torch)
    assert np.all(np.isfinite(y))
    # a host pointer handed to an async entry point is refused, not dereferenced on the GPU
    with pytest.raises(_lib.CfsHipError, match="device pointers"):
        _lib.check(lib.cfs_hip_sym_spmv_async(A._h, y.ctypes.data, x.ctypes.data, None))
    A.close()


def test_two_handles_with_different_windows_interleaved():
    """hipFuncAttributeMaxDynamicSharedMemorySize belongs to the kernel instantiation, not
    to a handle: two handles of one instantiation with different LDS windows"""
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import synth
    from oracle import oracle
    torch = _torch()
    n, rp, ci, va, _ = synth.generate("Flan_1565", 0.05)
    x = synth.make_x(n)
    A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(max_slots=4992, flags=FLAG_NO_CAL | FLAG_NO_REORDER))
    B = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(max_slots=2900, flags=FLAG_NO_CAL | FLAG_NO_REORDER))
    assert A.stats()["block_threads"] == B.stats()["block_threads"]
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    for _ in range(3):
        assert scaled_err(_spmv(A, x, torch), y_ld, absrow) <= 1e-12
        assert scaled_err(_spmv(B, x, torch), y_ld, absrow) <= 1e-12
    A.close()
    B.close()


def test_refusals_have_their_own_error_codes():
    import scipy.sparse as sp
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import _lib
    _torch()
    n = 600
    L = sp.lil_matrix((n, n))
    L[n - 1, :n - 1] = 1.0
    A = (L + L.T + sp.identity(n)).tocsr()
    A.sort_indices()
    rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    with pytest.raises(_lib.CfsHipError) as e:
        cfs.SymMatrix(n, rp, ci, A.data, options=cfs.make_options(max_slots=128, flags=FLAG_NO_REORDER))
    assert e.value.code == _lib.ERR_UNSUPPORTED  # the one code the C++ surface falls back on
    with pytest.raises(_lib.CfsHipError) as e:
        cfs.SymMatrix(n, rp, ci, A.data, options=cfs.make_options(block_threads=300))
    assert e.value.code == _lib.ERR_ARG


@pytest.mark.parametrize("name", ["unstruct", "unstruct_bfs"])
def test_unstructured_stand_in_parity(name):
    """random point cloud, 21 nearest neighbours, 3 dof per node: sibling rows are NOT
    perfect prefixes of one another, the natural order of `unstruct` has no locality"""
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import synth
    from oracle import oracle
    torch = _torch()
    n, rp, ci, va, low = synth.generate(name, 0.05)
    x = synth.make_x(n)
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    for flags in (0, FLAG_NO_REORDER):
        A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=flags | FLAG_NO_CAL))
        assert scaled_err(_spmv(A, x, torch), y_ld, absrow) <= 1e-12
        assert A.stats()["nnz_low"] == low
        A.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name,scale,flags", [("Flan_1565", 0.05, 0), ("Flan_1565", 0.05, FLAG_NO_REORDER),
                                              ("ldoor", 0.15, FLAG_HYB), ("pdb1HYS", 0.5, FLAG_HYB | FLAG_CLUSTER)])
def test_update_values_refreshes_the_schedule_on_the_device(name, scale, flags, dtype):
    """cfs_hip_sym_update_values_*: new numbers, same sparsity pattern, poured into the
    existing schedule by a device kernel (no second tune()); host and device pointers"""
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import synth
    from oracle import oracle
    torch = _torch()
    n, rp, ci, va, _ = synth.generate(name, scale)
    va = va.astype(dtype)
    x = np.random.default_rng(2).uniform(-1, 1, n).astype(dtype)
    A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=flags | FLAG_NO_CAL | cfs.FLAG_KEEP_VALUE_MAP))
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    assert scaled_err(_spmv(A, x, torch), y_ld, absrow) <= TOL[dtype]
    # a symmetric perturbation of every value: v -> v * f(min(i, j), max(i, j)) + g
    rows = np.repeat(np.arange(n), np.diff(rp))
    lo, hi = np.minimum(rows, ci), np.maximum(rows, ci)
    va2 = (va.astype(np.float64) * (0.5 + ((lo * 31 + hi * 17) % 13) / 13.0) + 0.125 * ((lo + hi) % 3)).astype(dtype)
    y2_ld, absrow2 = oracle.csr_spmv_ld(n, rp, ci, va2, x)
    A.update_values(va2)                                   # host pointer
    assert scaled_err(_spmv(A, x, torch), y2_ld, absrow2) <= TOL[dtype]
    A.update_values(torch.from_numpy(va).cuda())           # device pointer: back to the first values
    assert scaled_err(_spmv(A, x, torch), y_ld, absrow) <= TOL[dtype]
    A.close()
    B = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=flags | FLAG_NO_CAL))
    from cfs_spmv_amd import _lib
    with pytest.raises(_lib.CfsHipError, match="KEEP_VALUE_MAP"):
        B.update_values(va2)
    B.close()


def test_update_values_on_a_multi_device_handle_and_mirrored_shards():
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import synth
    from oracle import oracle
    torch = _torch()
    n, rp, ci, va, _ = synth.generate("Flan_1565", 0.03)
    x = synth.make_x(n)
    A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=cfs.FLAG_KEEP_VALUE_MAP), ngpus=3)
    va2 = va * 0.75
    A.update_values(va2)
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va2, x)
    assert scaled_err(_spmv(A, x, torch), y_ld, absrow) <= 1e-12
    A.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_deterministic_mirrored_shards(dtype):
    """the deterministic y window in the shard instantiation of the kernel (one-sided
    off-block slots): every rank's block is bit-reproducible and within tolerance"""
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import _lib, synth
    from oracle import oracle
    torch = _torch()
    n, rp, ci, va, _ = synth.generate("Flan_1565", 0.03)
    va = va.astype(dtype)
    x = np.random.default_rng(8).uniform(-1, 1, n).astype(dtype)
    xd = torch.from_numpy(x).cuda()
    rs = cfs.balanced_splits(n, rp, ci, 3)
    y = np.zeros(n, dtype=dtype)
    for r in range(3):
        A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=FLAG_DET | FLAG_NO_CAL),
                          row_splits=rs, rank=r)
        outs = []
        for k in range(3):
            yb = torch.full((int(rs[r + 1] - rs[r]),), float(k), dtype=xd.dtype, device="cuda")
            A.spmv_phases(yb, xd, None, 3)
            torch.cuda.synchronize()
            outs.append(yb.cpu().numpy())
        assert np.array_equal(outs[0].view(np.uint8), outs[1].view(np.uint8))
        assert np.array_equal(outs[0].view(np.uint8), outs[2].view(np.uint8))
        y[rs[r]:rs[r + 1]] = outs[0]
        if r == 0:  # a deterministic handle keeps no value map semantics: refused, not mangled
            with pytest.raises(_lib.CfsHipError):
                A.update_values(va)
        A.close()
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    assert scaled_err(y, y_ld, absrow) <= TOL[dtype]
