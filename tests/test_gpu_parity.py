"""GPU parity: the HIP tile kernel (through the C ABI) against the CPU oracle on
the same seeded inputs.  Tolerances: fp64 1e-12, fp32 1e-5 relative to
max(|y_ref|, sum_j |a_ij||x_j|); the reference's own
criterion (rel. 1e-8 / 1e-4 element-wise, include/utils/platform.hpp:27-37) is
asserted as well."""
import numpy as np
import pytest

from conftest import scaled_err

pytestmark = pytest.mark.gpu

TOL = {np.float64: 1e-12, np.float32: 1e-5}
REF_EPS = {np.float64: 1e-8, np.float32: 1e-4}


def _torch():
    import torch
    assert torch.cuda.is_available(), "gpu test without a GPU"
    return torch


def _gpu_spmv(A, x, torch, garbage=777.0):
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    yd = torch.full((A.nrows(),), garbage, dtype=xd.dtype, device="cuda")
    A.dense_vector_multiply(yd, xd)
    torch.cuda.synchronize()
    return yd.cpu().numpy()


def _check(n, rp, ci, va, x, dtype, options=None, threads=(1, 4), ref_criterion=True):
    import cfs_spmv_amd as cfs
    from oracle import oracle
    torch = _torch()
    va = va.astype(dtype)
    x = x.astype(dtype)
    A = cfs.SymMatrix(n, rp, ci, va, options=options)
    y = _gpu_spmv(A, x, torch)
    y2 = _gpu_spmv(A, x, torch, garbage=-3.25)  # twice: y must be re-initialised
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    tol = TOL[dtype]
    assert scaled_err(y, y_ld, absrow) <= tol
    assert scaled_err(y2, y_ld, absrow) <= tol
    for T in threads:
        o = oracle.SymOracle(n, rp, ci, va, T)
        y_ref = o.spmv(x)
        assert scaled_err(y, y_ref.astype(np.float64), absrow) <= tol, f"T={T}"
        o.close()
    # the reference's own test: SSS result vs plain CSR result, element-wise
    if ref_criterion:
        y_csr = oracle.csr_spmv(n, rp, ci, va, x)
        well = absrow <= 1e3 * np.abs(y_csr)  # rows that do not cancel
        assert np.all(np.abs(y[well] - y_csr[well]) <= REF_EPS[dtype] * np.abs(y[well]))
    A.close()
    return y


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name,scale", [("pdb1HYS", 0.25), ("pwtk", 0.05), ("ldoor", 0.02),
                                        ("Flan_1565", 0.01),
                                        # non-regular shapes: unequal unknowns
                                        # per node, rows that are not prefixes of each other; a
                                        # power-law graph with hub rows and long tails
                                        ("tetmesh", 0.03), ("powerlaw", 0.05)])
def test_synth_parity(name, scale, dtype):
    from cfs_spmv_amd import synth
    n, rp, ci, va, _ = synth.generate(name, scale)
    x = synth.make_x(n)
    # (the reference's element-wise isEqual compares against cpu_mv, which adds a row in the
    # working precision in stored order: on the hub rows of the power-law graph -- 16 000
    # entries -- that sum itself is only good to ~1e-3 in single precision, while the 1e-5 bound
    # against the long-double row sums above holds)
    _check(n, rp, ci, va, x, dtype, ref_criterion=not (name == "powerlaw" and dtype == np.float32))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,avg,band,slots,block", [
    (1, 0, None, 0, 0), (2, 1, None, 0, 0), (63, 3, None, 64, 256), (64, 3, None, 0, 256),
    (65, 3, None, 0, 512), (300, 5, None, 64, 256), (1000, 20, 50, 128, 1024),
    (5000, 8, None, 0, 0), (4097, 40, 300, 512, 512), (20000, 3, 2000, 2560, 256)])
def test_random_parity(n, avg, band, slots, block, dtype):
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import synth
    n, rp, ci, va = synth.random_symmetric(n, avg, seed=n * 7 + avg, band=band)
    rng = np.random.default_rng(n)
    x = rng.uniform(-1.0, 1.0, size=n)  # mixed signs: cancelling rows
    opt = cfs.make_options(max_slots=slots, block_threads=block)
    _check(n, rp, ci, va, x, dtype, options=opt, threads=(1, 2) if n >= 64 else (1,))


def test_ragged_and_empty_rows():
    """rows with no lower entries, a missing diagonal, duplicates-free arrow"""
    import scipy.sparse as sp
    n = 700
    rng = np.random.default_rng(3)
    L = sp.lil_matrix((n, n))
    for i in range(1, n):
        if i % 3 == 0:
            continue  # empty lower rows
        k = int(rng.integers(1, 90)) if i % 50 else min(i, 400)
        cols = np.unique(rng.integers(0, i, size=k))
        L[i, cols] = rng.uniform(-2, 2, size=cols.size)
    d = rng.uniform(1, 2, size=n)
    d[10] = 0.0
    A = (L + L.T + sp.diags(d)).tocsr()
    A.eliminate_zeros()  # row 10 has no stored diagonal now
    A.sort_indices()
    x = rng.uniform(-1, 1, size=n)
    _check(n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data, x, np.float64,
           threads=(1,))


def test_host_pointer_path_and_determinism():
    """cfs_hip_sym_spmv with host pointers (the unmodified-caller path of
    test/test_spmv_mmf.cpp) and run-to-run agreement"""
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import synth
    from oracle import oracle
    _torch()
    n, rp, ci, va, _ = synth.generate("pwtk", 0.05)
    x = synth.make_x(n)
    A = cfs.SymMatrix(n, rp, ci, va)
    f = cfs.SpDMV(A, cfs.Tuning.Aggressive)
    y = np.full(n, 9.0)
    f(y, n, x, n)
    f(y, n, x, n)
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    assert scaled_err(y, y_ld, absrow) <= 1e-12
    y3 = np.empty(n)
    f(y3, n, x, n)
    # LDS float atomics: the order of the transposed updates may vary run to run
    assert scaled_err(y3, y, absrow) <= 1e-13
    with pytest.raises(AssertionError):
        f(y, n + 1, x, n)


def test_csr_general_parity():
    """Format::csr on the GPU against cpu_mv_serial (csr_matrix.tpp:2664-2681)"""
    import scipy.sparse as sp
    import cfs_spmv_amd as cfs
    from oracle import oracle
    torch = _torch()
    for dtype in (np.float64, np.float32):
        for (m, k, dens) in [(1000, 1000, 0.01), (513, 700, 0.05), (4000, 4000, 0.002)]:
            A = sp.random(m, k, density=dens, random_state=5, format="csr", dtype=np.float64)
            A.sort_indices()
            rp, ci, va = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(dtype)
            x = np.random.default_rng(1).uniform(-1, 1, size=k).astype(dtype)
            G = cfs.CsrMatrix(m, k, rp, ci, va)
            xd = torch.from_numpy(x).cuda()
            yd = torch.full((m,), 5.0, dtype=xd.dtype, device="cuda")
            G.dense_vector_multiply(yd, xd)
            torch.cuda.synchronize()
            y = yd.cpu().numpy()
            y_ld, absrow = oracle.csr_spmv_ld(m, rp, ci, va, np.resize(x, k))
            assert scaled_err(y, y_ld, absrow) <= TOL[dtype]
            G.close()


@pytest.mark.parametrize("form", ["wave", "block"])
@pytest.mark.parametrize("name,scale", [("pdb1HYS", 0.25), ("pwtk", 0.05), ("ldoor", 0.02), ("Flan_1565", 0.01),
                                        ("Queen_4147", 0.005), ("tetmesh", 0.03), ("powerlaw", 0.05)])
def test_csr_general_parity_on_stand_ins(name, scale, form, monkeypatch):
    """both forms of the general CSR kernel (wave-stream: the default; workgroup-per-block) on
    the full CSR of the stand-ins -- short rows, 7-dof blocks, hub rows of > 10 000 entries
    (cfs_csr_longrow_kernel), empty tails -- against the long-double row sums"""
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import synth
    from oracle import oracle
    torch = _torch()
    monkeypatch.setenv("CFS_HIP_CSR_KERNEL", form)
    n, rp, ci, va, _ = synth.generate(name, scale)
    for dtype in (np.float64, np.float32):
        v = va.astype(dtype)
        x = synth.make_x(n, 42, dtype)
        G = cfs.CsrMatrix(n, n, rp, ci, v)
        xd = torch.from_numpy(x).cuda()
        yd = torch.full((n,), float("nan"), dtype=xd.dtype, device="cuda")
        G.dense_vector_multiply(yd, xd)
        G.dense_vector_multiply(yd, xd)
        torch.cuda.synchronize()
        y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, v, x)
        assert scaled_err(yd.cpu().numpy(), y_ld, absrow) <= TOL[dtype], (name, form, dtype.__name__)
        G.close()


@pytest.mark.parametrize("col16", ["1", "0"])
def test_csr_block_form_with_narrow_and_wide_row_blocks(col16, monkeypatch):
    """the block form keeps 16-bit column codes for row blocks whose columns fit four windows of
    16 384 and the 32-bit columns for the others: a matrix with both kinds of block (three bands
    40 000 columns apart -- three windows --, then rows with columns anywhere in 0 .. 300 000;
    odd row lengths so that blocks start at odd positions),
    with and without the 16-bit array, against the long-double row sums"""
    import scipy.sparse as sp
    import cfs_spmv_amd as cfs
    from oracle import oracle
    torch = _torch()
    monkeypatch.setenv("CFS_HIP_CSR_KERNEL", "block")
    monkeypatch.setenv("CFS_HIP_CSR_COL16", col16)
    rng = np.random.default_rng(5)
    n = 300_000
    ia = np.arange(200_000)  # stencil-like part: 5-9 entries within +-3000 columns of row - 40 000, row, row + 40 000
    ra = np.repeat(ia, 5 + ia % 5)
    ca = np.clip(ra + rng.choice([-40_000, 0, 40_000], ra.size) + rng.integers(-3000, 3000, ra.size), 0, n - 1)
    ib = np.arange(200_000, 260_000)  # wide part: 3-6 entries, columns anywhere
    rb = np.repeat(ib, 3 + ib % 4)
    cb = rng.integers(0, n, rb.size)
    r, c = np.concatenate([ra, rb]), np.concatenate([ca, cb])
    A = sp.csr_matrix((rng.standard_normal(r.size), (r, c)), shape=(n, n))
    A.sum_duplicates()
    A.sort_indices()
    rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    for dtype in (np.float64, np.float32):
        v = A.data.astype(dtype)
        x = rng.standard_normal(n).astype(dtype)
        G = cfs.CsrMatrix(n, n, rp, ci, v)
        xd = torch.from_numpy(x).cuda()
        yd = torch.full((n,), float("nan"), dtype=xd.dtype, device="cuda")
        G.dense_vector_multiply(yd, xd)
        torch.cuda.synchronize()
        y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, v, x)
        assert scaled_err(yd.cpu().numpy(), y_ld, absrow) <= TOL[dtype], (col16, dtype.__name__)
        G.close()


@pytest.mark.parametrize("last_col,narrow", [(65535, True), (65536, False)])
def test_csr_column_code_window_boundaries(last_col, narrow, monkeypatch):
    """the 16-bit column codes at their limits: columns 0, 16 383 | 16 384, 32 767 | 32 768, 49 151 |
    49 152, 65 535 fill four windows of 16 384 to the last offset; one column more (65 536) needs a fifth
    window and the block keeps its 32-bit columns.  cfs_hip_csr_stats says which, y is right either way"""
    import ctypes as C
    import scipy.sparse as sp
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import _lib
    from oracle import oracle
    torch = _torch()
    monkeypatch.setenv("CFS_HIP_CSR_KERNEL", "block")
    n = 70_000
    cols = np.array([0, 16383, 16384, 32767, 32768, 49151, 49152, last_col])
    nrows_used = 40  # one block
    r = np.repeat(np.arange(nrows_used), cols.size)
    c = np.tile(cols, nrows_used)
    rng = np.random.default_rng(9)
    A = sp.csr_matrix((rng.standard_normal(r.size), (r, c)), shape=(n, n))
    rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    for dtype in (np.float64, np.float32):
        v = A.data.astype(dtype)
        x = rng.standard_normal(n).astype(dtype)
        G = cfs.CsrMatrix(n, n, rp, ci, v)
        streamed, nar = C.c_int64(), C.c_int64()
        _lib.check(_lib.load().cfs_hip_csr_stats(G._h, C.byref(streamed), C.byref(nar)))
        assert (nar.value == A.nnz) == narrow and nar.value in (0, A.nnz)
        xd = torch.from_numpy(x).cuda()
        yd = torch.full((n,), float("nan"), dtype=xd.dtype, device="cuda")
        G.dense_vector_multiply(yd, xd)
        torch.cuda.synchronize()
        y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, v, x)
        assert scaled_err(yd.cpu().numpy(), y_ld, absrow) <= TOL[dtype]
        G.close()


@pytest.mark.parametrize("nranks", [2, 4])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_shards_on_one_device(nranks, dtype):
    """1-D row-block shards, all on cuda:0: local tile kernels, pack of
    the off-block contributions, routing by the static row lists (what the RCCL
    all-to-all does between processes), owner-side fold"""
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import synth
    from oracle import oracle
    torch = _torch()
    n, rp, ci, va, _ = synth.generate("Flan_1565", 0.02)
    va = va.astype(dtype)
    x = synth.make_x(n, 42, dtype)
    rs = cfs.balanced_splits(n, rp, ci, nranks)
    xd = torch.from_numpy(x).cuda()
    tdt = xd.dtype
    xopt = cfs.make_options(flags=cfs.FLAG_SHARD_EXCHANGE)  # the exchange form of a shard
    shards = [cfs.SymMatrix(n, rp, ci, va, options=xopt, row_splits=rs, rank=r)
              for r in range(nranks)]
    send_rows = [s.send_rows() for s in shards]
    send_counts = [s.send_counts() for s in shards]
    # receive lists: concatenation by source rank of the rows aimed at me
    recv_rows = []
    for r in range(nranks):
        parts = []
        for src in range(nranks):
            off = int(send_counts[src][:r].sum())
            parts.append(send_rows[src][off:off + int(send_counts[src][r])])
        recv_rows.append(np.concatenate(parts) if parts else np.zeros(0, np.int32))
        shards[r].set_recv(recv_rows[r])
    y = np.zeros(n, dtype=dtype)
    send_bufs, yb = [], []
    for r, s in enumerate(shards):
        yb.append(torch.full((int(rs[r + 1] - rs[r]),), 3.0, dtype=tdt, device="cuda"))
        send_bufs.append(torch.zeros(max(1, send_rows[r].size), dtype=tdt, device="cuda"))
        s.spmv_local(yb[r], xd, send_bufs[r])
    torch.cuda.synchronize()
    for r, s in enumerate(shards):
        parts = []
        for src in range(nranks):
            off = int(send_counts[src][:r].sum())
            parts.append(send_bufs[src][off:off + int(send_counts[src][r])])
        recv = torch.cat(parts) if recv_rows[r].size else torch.zeros(1, dtype=tdt, device="cuda")
        s.recv_fold(yb[r], recv)
        torch.cuda.synchronize()
        y[rs[r]:rs[r + 1]] = yb[r].cpu().numpy()
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    assert scaled_err(y, y_ld, absrow) <= TOL[dtype]
    for s in shards:
        s.close()


@pytest.mark.parametrize("flags", [0, 8])
def test_long_rows_are_split_over_lanes(flags):
    """a few rows 30x longer than the rest: virtual rows (chunks of one row on
    several lanes, partial sums meeting in one LDS slot), many slices per wave
    (ticket scheduling), both row orders"""
    import scipy.sparse as sp
    import cfs_spmv_amd as cfs
    n = 6000
    rng = np.random.default_rng(11)
    rows, cols = [], []
    for i in range(1, n):
        k = 600 if (i % 997 == 0) else 12
        c = np.unique(rng.integers(max(0, i - 1500), i, size=min(k, i)))
        rows.append(np.full(c.size, i))
        cols.append(c)
    r, c = np.concatenate(rows), np.concatenate(cols)
    L = sp.coo_matrix((rng.uniform(-1, 1, r.size), (r, c)), shape=(n, n)).tocsr()
    A = (L + L.T + sp.diags(rng.uniform(1, 2, n))).tocsr()
    A.sort_indices()
    x = rng.uniform(-1, 1, n)
    for dtype in (np.float64, np.float32):
        _check(n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data, x, dtype,
               options=cfs.make_options(max_slots=2496, flags=flags), threads=(1, 3))


def test_natural_and_clustered_orders_agree():
    """CFS_HIP_FLAG_NO_REORDER vs the default on a matrix where clustering is
    chosen (3-D stencil): same y up to summation order"""
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import synth
    from oracle import oracle
    torch = _torch()
    n, rp, ci, va, _ = synth.generate("Flan_1565", 0.05)
    x = synth.make_x(n)
    ys = []
    for flags in (0, 8):
        A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=flags))
        ys.append(_gpu_spmv(A, x, torch))
        A.close()
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    assert scaled_err(ys[0], y_ld, absrow) <= 1e-12 and scaled_err(ys[1], y_ld, absrow) <= 1e-12
    assert scaled_err(ys[0], ys[1], absrow) <= 1e-13


@pytest.mark.parametrize("name,scale,nranks,flags", [
    ("Flan_1565", 0.02, 2, 0), ("Flan_1565", 0.02, 4, 0), ("Flan_1565", 0.05, 3, 8),
    ("ldoor", 0.03, 4, 0), ("pwtk", 0.1, 8, 0)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_mirrored_shards_need_no_exchange(name, scale, nranks, flags, dtype):
    """default form of a shard: off-block entries are stored by both ranks they touch
    and processed one-sided -- every rank's block of y is complete after its own
    local launches, nothing is packed, sent or received"""
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import synth
    from oracle import oracle
    torch = _torch()
    n, rp, ci, va, low = synth.generate(name, scale)
    va = va.astype(dtype)
    x = synth.make_x(n, 42, dtype)
    rs = cfs.balanced_splits(n, rp, ci, nranks)
    xd = torch.from_numpy(x).cuda()
    y = np.zeros(n, dtype=dtype)
    mirrored = 0
    for r in range(nranks):
        A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=flags), row_splits=rs, rank=r)
        st = A.stats()
        assert st["remote_vals"] == 0 and A.send_rows().size == 0
        mirrored += st["mirror_entries"]
        yb = torch.full((int(rs[r + 1] - rs[r]),), float("nan"), dtype=xd.dtype, device="cuda")
        A.spmv_phases(yb, xd, None, 7)
        torch.cuda.synchronize()
        y[rs[r]:rs[r + 1]] = yb.cpu().numpy()
        A.close()
    rows_of = np.repeat(np.arange(n), np.diff(rp))
    owner = np.searchsorted(rs, rows_of, side="right") - 1
    col_owner = np.searchsorted(rs, ci, side="right") - 1
    assert mirrored == int(np.sum(col_owner < owner)) > 0  # each off-block lower entry once more
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    assert scaled_err(y, y_ld, absrow) <= TOL[dtype]


def test_cg_solver_loop_on_the_gpu():
    """solver-style caller: conjugate gradients whose every product is
    the HIP path on resident vectors; the answer is checked against a direct solve"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import synth
    from cfs_spmv_amd.solver import cg
    torch = _torch()
    n, rp, ci, va, _ = synth.generate("pwtk", 0.05)
    A = cfs.SymMatrix(n, rp, ci, va)
    b = synth.make_x(n, 11)
    bd = torch.from_numpy(b).cuda()
    # the asynchronous entry point enqueues on torch's current stream, like the dots
    u, it, res = cg(A, bd, tol=1e-11, maxiter=500)
    torch.cuda.synchronize()
    u_ref = spl.spsolve(sp.csc_matrix(sp.csr_matrix((va, ci, rp), shape=(n, n))), b)
    assert 0 < it < 500 and res <= 1e-10
    assert np.max(np.abs(u.cpu().numpy() - u_ref)) <= 1e-9 * np.max(np.abs(u_ref))
    A.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_native_cg_matches_the_host_driven_loop(dtype):
    """cfs_hip_sym_cg: the whole iteration behind the C ABI (five launches, scalars in device memory,
    the host looks at a flag every few iterations) against the torch-driven loop and a direct solve"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import synth
    from cfs_spmv_amd.solver import cg, cg_native
    torch = _torch()
    n, rp, ci, va, _ = synth.generate("pwtk", 0.05)
    va = va.astype(dtype)
    A = cfs.SymMatrix(n, rp, ci, va)
    b = synth.make_x(n, 11, dtype)
    bd = torch.from_numpy(b).cuda()
    tol = 1e-11 if dtype == np.float64 else 2e-5
    u1, it1, res1 = cg(A, bd, tol=tol, maxiter=500)
    for check_every in (1, 8, 1000):
        u2, it2, res2 = cg_native(A, bd, tol=tol, maxiter=500, check_every=check_every)
        torch.cuda.synchronize()
        assert 0 < it2 < 500 and abs(it2 - it1) <= 2, (it1, it2, check_every)
        assert res2 <= 10 * tol, (res1, res2)
    u_ref = spl.spsolve(sp.csc_matrix(sp.csr_matrix((va.astype(np.float64), ci, rp), shape=(n, n))), b.astype(np.float64))
    lim = 1e-9 if dtype == np.float64 else 2e-3
    assert np.max(np.abs(u2.cpu().numpy() - u_ref)) <= lim * np.max(np.abs(u_ref))
    # the iteration limit: stops there, reports the residual it reached
    u3, it3, res3 = cg_native(A, bd, tol=tol, maxiter=3)
    assert it3 == 3 and res3 > tol
    # a first guess that already solves the system: no iteration
    u4, it4, res4 = cg_native(A, bd, tol=tol, maxiter=500, x0=u2)
    assert it4 <= 1 and res4 <= 10 * tol
    A.close()
    # a multi-device handle (two shards, here on one device): the products on the shards' streams,
    # the vector kernels on the caller's
    M = cfs.SymMatrix(n, rp, ci, va, ngpus=2)
    um, itm, resm = cg_native(M, bd, tol=tol, maxiter=500)
    torch.cuda.synchronize()
    assert abs(itm - it1) <= 2 and resm <= 10 * tol
    assert np.max(np.abs(um.cpu().numpy() - u_ref)) <= lim * np.max(np.abs(u_ref))
    M.close()
    # with a deterministic handle the whole solve is bit-reproducible: the scalars are sums of
    # per-workgroup partial sums in a fixed order, not atomics
    D = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=1024))
    ua, ita, _ = cg_native(D, bd, tol=tol, maxiter=60)
    ub, itb, _ = cg_native(D, bd, tol=tol, maxiter=60)
    torch.cuda.synchronize()
    assert ita == itb and torch.equal(ua, ub)
    D.close()


def _fallback_worker(rank, world, port, q):
    try:
        import os, sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import scipy.sparse as sp
        import torch
        import torch.distributed as dist
        import cfs_spmv_amd as cfs
        from cfs_spmv_amd import synth
        from cfs_spmv_amd.dist import build_shard
        from oracle import oracle
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        n, rp, ci, va, _ = synth.generate("pwtk", 0.03)
        rs = cfs.balanced_splits(n, rp, ci, world)
        res = []
        for broken in (False, True):
            if broken:  # drop the upper image of one lower entry across the cut
                rows = np.repeat(np.arange(n), np.diff(rp))
                k = int(np.flatnonzero((rows >= rs[1]) & (ci < rs[1]))[0])
                A = sp.csr_matrix((va, ci, rp), shape=(n, n)).tolil()
                A[int(ci[k]), int(rows[k])] = 0
                A = A.tocsr()
                A.eliminate_zeros()
                A.sort_indices()
                rp2, ci2, va2 = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data
            else:
                rp2, ci2, va2 = rp, ci, va
            M, sh, used = build_shard(n, rp2, ci2, va2, world, rank, rs, torch.device("cuda", 0),
                                      exchange="none", stage_via_host=True)
            x = synth.make_x(n)
            xd = torch.from_numpy(x).cuda()
            yb = torch.full((int(rs[rank + 1] - rs[rank]),), float("nan"), dtype=torch.float64,
                            device="cuda")
            sh.spmv(yb, xd)
            torch.cuda.synchronize()
            # the reference semantics: the LOWER triangle defines the operator
            L = sp.tril(sp.csr_matrix((va2, ci2, rp2), shape=(n, n)), -1)
            S = (L + L.T + sp.diags(sp.csr_matrix((va2, ci2, rp2), shape=(n, n)).diagonal())).tocsr()
            S.sort_indices()
            y_ld, absrow = oracle.csr_spmv_ld(n, S.indptr.astype(np.int32),
                                              S.indices.astype(np.int32), S.data, x)
            sl = slice(int(rs[rank]), int(rs[rank + 1]))
            err = float(np.max(np.abs(yb.cpu().numpy() - y_ld[sl]) /
                               np.maximum(np.abs(y_ld[sl]), absrow[sl])))
            res.append((used, err))
            M.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, res))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, f"{e}\n{traceback.format_exc()}"))


def test_build_shard_forms_agree_across_ranks():
    """two ranks sharing cuda:0 over gloo: a mirrorable matrix runs without exchange; one
    whose off-block structure is unsymmetric makes rank 0 fail to mirror, and BOTH ranks
    fall back to the exchange form together -- with the right answer either way"""
    import socket
    import torch.multiprocessing as mp
    _torch()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    for rank, res in out:
        assert not isinstance(res, str), f"rank {rank}: {res}"
        (used0, err0), (used1, err1) = res
        assert used0 == "none" and used1 == "all_to_all", res
        assert err0 <= 1e-12 and err1 <= 1e-12, res


@pytest.mark.parametrize("seed", range(24))
def test_random_shapes_parity(seed):
    """the shapes of tests/test_plan_random.py through the kernels: banded, random, node
    blocks, hub rows; random schedule options; whole matrix and mirrored shards"""
    import cfs_spmv_amd as cfs
    from cfs_spmv_amd import _lib
    from oracle import oracle
    from rand_matrices import random_matrix
    torch = _torch()
    rng = np.random.default_rng(5000 + seed)
    kind = ["band", "random", "nodes", "hub"][seed % 4]
    n, A = random_matrix(rng, int(rng.integers(300, 6000)), kind)
    dtype = np.float64 if seed % 3 else np.float32
    rp, ci, va = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(dtype)
    x = rng.uniform(-1, 1, n).astype(dtype)
    opt = cfs.make_options(max_slots=int(rng.choice([0, 512, 2560])),
                           block_threads=int(rng.choice([256, 512, 1024])),
                           flags=int(rng.choice([0, 0, 8, 16])))
    nranks = int(rng.choice([1, 1, 2, 4]))
    rs = cfs.balanced_splits(n, rp, ci, nranks) if nranks > 1 else np.array([0, n], dtype=np.int32)
    xd = torch.from_numpy(x).cuda()
    y = np.zeros(n, dtype=dtype)
    for r in range(nranks):
        try:
            M = cfs.SymMatrix(n, rp, ci, va, options=opt,
                              row_splits=rs if nranks > 1 else None, rank=r)
        except _lib.CfsHipError as e:
            assert "dense row" in str(e), str(e)
            return
        yb = torch.full((int(rs[r + 1] - rs[r]),), float("nan"), dtype=xd.dtype, device="cuda")
        M.spmv_phases(yb, xd, None, 7)
        torch.cuda.synchronize()
        y[rs[r]:rs[r + 1]] = yb.cpu().numpy()
        M.close()
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    assert scaled_err(y, y_ld, absrow) <= TOL[dtype]
