"""tune() on the GPU (cfs_devplan.hpp): the schedule the HIP kernels build from
one upload of the caller's CSR is BIT-IDENTICAL to the one the host builder (cfs_plan.hpp)
uploads -- tiles, slot tables, virtual rows, slice metadata, leader lanes, packet streams, COO
leftovers, fold records -- on the stand-ins, on randomised matrices of many shapes and option
sets (the cases of test_plan_random.py) and on mirrored shards; and its SpMV agrees with the
oracle.  The reference's counterpart of this work is the host-side preprocessing of
conflict_free_aposteriori, include/matrix/csr_matrix.tpp:1204-1639."""
import numpy as np
import pytest

import cfs_spmv_amd as cfs
from cfs_spmv_amd import _lib, synth
from rand_matrices import random_matrix

pytestmark = pytest.mark.gpu

NO_CALIBRATE = 32


@pytest.fixture(autouse=True)
def _torch_first():
    """torch brings a HIP runtime of its own: it has to initialise before libcfs_hip.so's"""
    import torch
    torch.cuda.init()
    torch.cuda.set_device(0)
    yield


def _both(n, rp, ci, va, flags=0, slots=0, block=0, row_splits=None, rank=0):
    """(device-built handle, host-built handle) for the same matrix and options"""
    kw = dict(row_splits=row_splits, rank=rank)
    D = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(slots, 0, block, flags | NO_CALIBRATE), **kw)
    H = cfs.SymMatrix(n, rp, ci, va,
                      options=cfs.make_options(slots, 0, block, flags | NO_CALIBRATE | cfs.FLAG_HOST_PLAN), **kw)
    return D, H


def _assert_same(D, H, what):
    d, h = D.digest(), H.digest()
    assert h["device_built"] == 0
    assert d["device_built"] == 1, f"{what}: the device builder handed over to the host builder: {D.plan_note()}"
    diff = [k for k in d if k != "device_built" and d[k] != h[k]]
    sd, sh = D.stats(), H.stats()
    assert not diff, (what, diff, {k: (sd[k], sh[k]) for k in sd if sd[k] != sh[k]})
    for k in ("ntiles", "nslices", "halo_slots", "fold_rows", "bytes_streamed", "lds_bytes", "ngroups",
              "mirror_entries", "nnz_low", "nnz_full"):
        assert sd[k] == sh[k], (what, k, sd[k], sh[k])


@pytest.mark.parametrize("name,scale", [("pwtk", 0.05), ("pwtk", 1.0), ("ldoor", 0.05), ("Flan_1565", 0.03),
                                        ("Flan_1565", 0.3), ("pdb1HYS", 0.3), ("unstruct", 0.05)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_device_schedule_equals_host_schedule_on_stand_ins(name, scale, dtype):
    n, rp, ci, va, _ = synth.generate(name, scale)
    va = va.astype(dtype)
    D, H = _both(n, rp, ci, va)
    _assert_same(D, H, (name, scale, dtype.__name__))
    # ... and it computes the same y (same arrays, same kernel: bit for bit in one launch order)
    import torch
    x = torch.from_numpy(synth.make_x(n, 42, dtype)).cuda()
    yd = torch.empty(n, dtype=x.dtype, device="cuda")
    yh = torch.empty_like(yd)
    D.dense_vector_multiply(yd, x)
    H.dense_vector_multiply(yh, x)
    torch.cuda.synchronize()
    tol = 1e-12 if dtype == np.float64 else 1e-5
    scale_ = torch.maximum(yh.abs(), torch.tensor(1.0, dtype=x.dtype, device="cuda"))
    assert float(((yd - yh).abs() / scale_).max()) <= 10 * tol
    D.close()
    H.close()


FLAG_HYB, FLAG_NO_REORDER, FLAG_CLUSTER, FLAG_DET = 128, 8, 16, 1024


@pytest.mark.parametrize("name,scale,flags,slots", [
    ("ldoor", 0.05, 0, 0), ("ldoor", 0.3, 0, 0), ("ldoor", 0.1, FLAG_NO_REORDER, 0), ("ldoor", 0.1, FLAG_CLUSTER, 1024),
    ("unstruct", 0.05, FLAG_NO_REORDER, 0), ("unstruct", 0.05, 0, 0), ("pwtk", 0.2, 0, 512), ("Flan_1565", 0.05, 0, 1024),
    ("pdb1HYS", 0.3, FLAG_NO_REORDER, 0), ("powerlaw", 0.02, 0, 0), ("ldoor", 0.1, cfs.FLAG_KEEP_VALUE_MAP, 0),
    ("ldoor", 0.1, FLAG_DET, 0)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_device_schedule_with_far_entries(name, scale, flags, slots, dtype):
    """Format::hyb (CFS_HIP_FLAG_HYB): far marks per tile, the second and third cut without them, the
    marks resolved against the final tiles, the near entries compacted for the tile format and the
    far sections (own rows' entries, then the mirror images sorted by row and column) -- on the
    device as on the host, array for array"""
    import torch
    n, rp, ci, va, _ = synth.generate(name, scale)
    va = va.astype(dtype)
    D, H = _both(n, rp, ci, va, flags=flags | FLAG_HYB, slots=slots)
    _assert_same(D, H, ("hyb", name, scale, flags, slots, dtype.__name__))
    sd, sh = D.stats(), H.stats()
    assert sd["far_entries"] == sh["far_entries"]
    if name in ("ldoor", "unstruct"):
        assert sd["far_entries"] > 0 and D.digest()["fcols"] != 0
    x = torch.from_numpy(synth.make_x(n, 42, dtype)).cuda()
    yd, yh = torch.empty(n, dtype=x.dtype, device="cuda"), torch.empty(n, dtype=x.dtype, device="cuda")
    D.dense_vector_multiply(yd, x)
    H.dense_vector_multiply(yh, x)
    torch.cuda.synchronize()
    if flags & FLAG_DET:
        assert torch.equal(yd, yh)
    else:
        tol = 1e-12 if dtype == np.float64 else 1e-5
        scale_ = torch.maximum(yh.abs(), torch.tensor(1.0, dtype=x.dtype, device="cuda"))
        assert float(((yd - yh).abs() / scale_).max()) <= 10 * tol
    D.close()
    H.close()


@pytest.mark.parametrize("nranks", [2, 3])
def test_device_schedule_of_mirrored_shards_with_far_entries(nranks):
    n, rp, ci, va, _ = synth.generate("ldoor", 0.1)
    rs = cfs.balanced_splits(n, rp, ci, nranks)
    for rank in range(nranks):
        D, H = _both(n, rp, ci, va, flags=FLAG_HYB, row_splits=rs, rank=rank)
        _assert_same(D, H, ("hyb mirrored shard", nranks, rank))
        assert D.stats()["far_entries"] == H.stats()["far_entries"] > 0
        D.close()
        H.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_tuned_alternative_reuses_the_kept_placement(dtype, monkeypatch):
    """Tuning::Aggressive builds a second schedule (1 024 threads x 1 per CU, half as many clusters:
    pairs of the first schedule's) -- on the device from the KEPT upload and placement of the first
    build (cfs_dev::Kept), on the host from its schedule-space matrix: the same schedule again"""
    monkeypatch.setenv("CFS_HIP_SHAPE", "1024")  # keep the alternative whatever the clock says
    n, rp, ci, va, _ = synth.generate("Flan_1565", 0.3)
    va = va.astype(dtype)
    D = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=16))  # clustered order
    H = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=16 | cfs.FLAG_HOST_PLAN))
    assert D.stats()["block_threads"] == 1024 and H.stats()["block_threads"] == 1024
    _assert_same(D, H, ("tuned alternative", dtype.__name__))
    D.close()
    H.close()


@pytest.mark.parametrize("name,scale,shape", [("ldoor", 0.3, "512"), ("ldoor", 0.3, "1024"), ("unstruct", 0.1, "512"),
                                              ("unstruct", 0.1, "1024")])
def test_tuned_hyb_alternative_reuses_upload_and_clusters(name, scale, shape, monkeypatch):
    """Tuning::Aggressive may build three schedules of one matrix: two window shapes and Format::hyb
    for the kept one.  On the device the later builds read the KEPT upload, and the HYB build of
    the kept shape also the clusters (and their placement) that shape was built from -- whichever
    row order won.  The choices are pinned (the clock decides otherwise): the result equals the
    host builder's, array for array"""
    monkeypatch.setenv("CFS_HIP_SHAPE", shape)
    monkeypatch.setenv("CFS_HIP_TAKE_HYB", "1")
    n, rp, ci, va, _ = synth.generate(name, scale)
    D = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options())
    H = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=cfs.FLAG_HOST_PLAN))
    assert D.stats()["block_threads"] == int(shape) == H.stats()["block_threads"]
    assert D.stats()["far_entries"] == H.stats()["far_entries"]
    if name == "ldoor":  # (the clustered stand-in has too few single-use halo columns: HYB is not tried)
        assert D.stats()["far_entries"] > 0
    _assert_same(D, H, ("tuned + hyb", name, shape))
    D.close()
    H.close()


@pytest.mark.parametrize("seed", range(48))
def test_device_schedule_equals_host_schedule_on_random_matrices(seed):
    rng = np.random.default_rng(5000 + seed)
    kind = ["band", "random", "nodes", "hub"][seed % 4]
    n = int(rng.integers(300, 6000))
    n, A = random_matrix(rng, n, kind)
    rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    va = A.data.astype(np.float64 if seed % 3 else np.float32)
    block = int(rng.choice([0, 256, 512, 1024]))
    slots = int(rng.choice([0, 256, 1024, 2560]))
    flags = int(rng.choice([0, 0, 8, 16]))  # default order choice / natural / clustered
    if seed % 5 == 0:
        flags |= cfs.FLAG_KEEP_VALUE_MAP
    if seed % 3 == 1:
        flags |= FLAG_HYB
    try:
        D, H = _both(n, rp, ci, va, flags, slots, block)
    except _lib.CfsHipError as e:
        assert "dense row" in str(e), str(e)
        return
    if D.digest()["device_built"] == 0:  # long / dense rows: the host builder took it
        assert kind == "hub", (kind, n, block, slots, flags, D.plan_note())
        return
    _assert_same(D, H, (kind, n, block, slots, flags))
    D.close()
    H.close()


@pytest.mark.parametrize("name,scale,dtype", [("pwtk", 0.1, np.float64), ("Flan_1565", 0.05, np.float32),
                                              ("tetmesh", 0.05, np.float64)])
def test_device_schedule_of_the_deterministic_build(name, scale, dtype):
    """CFS_HIP_FLAG_DETERMINISTIC: the smaller windows (26 bytes of LDS per slot) and the per-slot
    scale exponents come out of the device builder as out of the host builder; bit-identical y"""
    import torch
    n, rp, ci, va, _ = synth.generate(name, scale)
    va = va.astype(dtype)
    D, H = _both(n, rp, ci, va, flags=1024)
    _assert_same(D, H, ("deterministic", name))
    assert D.digest()["slot_exp"] != 0
    x = torch.from_numpy(synth.make_x(n, 42, dtype)).cuda()
    yd, yh = torch.empty(n, dtype=x.dtype, device="cuda"), torch.empty(n, dtype=x.dtype, device="cuda")
    D.dense_vector_multiply(yd, x)
    H.dense_vector_multiply(yh, x)
    torch.cuda.synchronize()
    assert torch.equal(yd, yh)  # deterministic mode: the same bits from both handles
    D.close()
    H.close()


@pytest.mark.parametrize("nranks", [2, 4])
def test_device_schedule_of_exchange_form_shards(nranks):
    """CFS_HIP_FLAG_SHARD_EXCHANGE: halo columns left of the block are SENT -- the send lists
    (rows, pointers, strip indices, counts per owner) of the device builder equal the host's"""
    n, rp, ci, va, _ = synth.generate("Flan_1565", 0.05)
    rs = cfs.balanced_splits(n, rp, ci, nranks)
    for rank in range(nranks):
        D, H = _both(n, rp, ci, va, flags=cfs.FLAG_SHARD_EXCHANGE, row_splits=rs, rank=rank)
        _assert_same(D, H, ("exchange shard", nranks, rank))
        assert np.array_equal(D.send_counts(), H.send_counts())
        assert np.array_equal(D.send_rows(), H.send_rows())
        assert D.stats()["remote_vals"] == H.stats()["remote_vals"]
        assert (D.stats()["remote_vals"] > 0) == (rank > 0)
        D.close()
        H.close()


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_device_schedule_of_mirrored_shards(nranks):
    import torch
    from oracle import oracle
    n, rp, ci, va, _ = synth.generate("Flan_1565", 0.05)
    rs = cfs.balanced_splits(n, rp, ci, nranks)
    x_host = synth.make_x(n)
    x = torch.from_numpy(x_host).cuda()
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x_host)
    for rank in range(nranks):
        D, H = _both(n, rp, ci, va, row_splits=rs, rank=rank)
        _assert_same(D, H, ("shard", nranks, rank))
        rows = int(rs[rank + 1] - rs[rank])
        y = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
        D.spmv_phases(y, x, None, 3)
        torch.cuda.synchronize()
        ref = y_ld[rs[rank]:rs[rank + 1]]
        den = np.maximum(np.abs(ref), absrow[rs[rank]:rs[rank + 1]])
        assert float(np.max(np.abs(y.cpu().numpy() - ref) / den)) <= 1e-12
        D.close()
        H.close()


def test_device_builder_hands_unsorted_rows_to_the_host_builder():
    """rows whose columns do not ascend are outside the device builder's contract: it must
    notice (on the GPU) and hand over, never build a wrong schedule"""
    import torch
    from oracle import oracle
    n, rp, ci, va, _ = synth.generate("pwtk", 0.02)
    ci, va = ci.copy(), va.copy()
    for i in range(0, n, 7):  # reverse every 7th row
        b, e = rp[i], rp[i + 1]
        ci[b:e] = ci[b:e][::-1]
        va[b:e] = va[b:e][::-1]
    A = cfs.SymMatrix(n, rp, ci, va, options=cfs.make_options(flags=NO_CALIBRATE))
    assert A.digest()["device_built"] == 0
    x_host = synth.make_x(n)
    x = torch.from_numpy(x_host).cuda()
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    A.dense_vector_multiply(y, x)
    torch.cuda.synchronize()
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x_host)
    assert float(np.max(np.abs(y.cpu().numpy() - y_ld) / np.maximum(np.abs(y_ld), absrow))) <= 1e-12
    A.close()
