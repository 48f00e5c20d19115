"""CPU-side checks of the product library (no compute calls without a GPU):
the C-ABI library loads and exports every symbol include/cfs_hip.h declares, and
the host-side tile schedule (what tune() uploads) encodes exactly the input."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import cfs_spmv_amd as cfs
from cfs_spmv_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "cfs_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cfs_hip_\w+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = cfs.load()
    assert lib.cfs_hip_abi_version() == 4
    syms = header_symbols()
    assert len(syms) >= 30
    for name in syms:
        assert hasattr(lib, name), f"{name} declared in cfs_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == syms, "cfs_spmv_amd/_lib.py SYMBOLS out of date"
    # raw dlsym as well (no ctypes caching involved)
    raw = C.CDLL(cfs.lib_path())
    for name in syms:
        getattr(raw, name)


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "HERE", str(tmp_path))
    with pytest.raises(_lib.CfsHipError, match="no CPU fallback"):
        _lib.load()


def test_bad_arguments_return_error_codes():
    lib = cfs.load()
    assert lib.cfs_hip_sym_destroy(None) == 0
    rep = _lib.PlanReport()
    rc = lib.cfs_hip_sym_plan_check_f64(4, None, None, None, 1, 0, None, None, C.byref(rep))
    assert rc != 0
    with pytest.raises(_lib.CfsHipError):
        _lib.check(rc)
    opt = cfs.make_options(block_threads=300)
    n, rp, ci, va = synth.random_symmetric(50, 3, 1)
    with pytest.raises(_lib.CfsHipError, match="block_threads"):
        cfs.plan_check(n, rp, ci, va, options=opt)


CASES = [("pdb1HYS", 0.1), ("pwtk", 0.05), ("ldoor", 0.02), ("Flan_1565", 0.02),
         ("Queen_4147", 0.005)]


@pytest.mark.parametrize("name,scale", CASES)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_schedule_encodes_the_lower_triangle(name, scale, dtype):
    n, rp, ci, va, low = synth.generate(name, scale)
    rep = cfs.plan_check(n, rp, ci, va.astype(dtype))
    assert rep["mismatches"] == 0
    assert rep["decoded"] == rep["nnz_low"] == low
    assert rep["ngroups"] % 8 == 0 and rep["ntiles"] >= 1
    assert rep["stream_len"] <= low + 8 * rep["nslices"] + 8 * rep["ntiles"]  # no padding entries


FLAG_HYB, FLAG_CLUSTER, FLAG_NO_REORDER, FLAG_EXCHANGE = 128, 16, 8, 64


@pytest.mark.parametrize("flags", [FLAG_CLUSTER, FLAG_NO_REORDER])
def test_full_grid_schedule(flags):
    """enough rows for one group per resident workgroup (512): the clusters grow to equal
    MODEL cost (tiles, halo slots) and the launch order is a permutation inside every
    XCD's run of slots -- plan_check verifies both the decoded entries and that order"""
    n, rp, ci, va, low = synth.generate("Flan_1565", 0.2)
    rep = cfs.plan_check(n, rp, ci, va, options=cfs.make_options(flags=flags))
    assert rep["mismatches"] == 0 and rep["decoded"] == low
    assert rep["ngroups"] == 512


@pytest.mark.parametrize("name,scale", CASES + [("Flan_1565", 0.2), ("ldoor", 0.3)])
@pytest.mark.parametrize("flags", [FLAG_HYB, FLAG_HYB | FLAG_CLUSTER, FLAG_HYB | FLAG_NO_REORDER])
def test_hyb_schedules_encode_the_lower_triangle(name, scale, flags):
    """Format::hyb (far entries kept by both tiles they touch, one-sided): the decoded
    triples are still exactly the strict lower triangle and every far entry has exactly
    one mirror image (checked inside cfs_hip_sym_plan_check_*)"""
    n, rp, ci, va, low = synth.generate(name, scale)
    rep = cfs.plan_check(n, rp, ci, va, options=cfs.make_options(flags=flags))
    assert rep["mismatches"] == 0 and rep["decoded"] == rep["nnz_low"] == low
    base = cfs.plan_check(n, rp, ci, va, options=cfs.make_options(flags=flags & ~FLAG_HYB))
    assert base["far_entries"] == 0
    # (the rows are re-cut with the far entries priced in: a few slots either way)
    assert rep["far_entries"] >= 0 and rep["halo_slots"] <= 1.02 * base["halo_slots"] + 64


@pytest.mark.parametrize("flags", [0, FLAG_NO_REORDER, FLAG_HYB, FLAG_CLUSTER | FLAG_HYB])
@pytest.mark.parametrize("name,scale", [("Flan_1565", 0.05), ("ldoor", 0.1), ("pdb1HYS", 0.3)])
def test_value_map_points_at_the_callers_values(name, scale, flags):
    """CFS_HIP_FLAG_KEEP_VALUE_MAP (2048): cfs_hip_sym_plan_check_* verifies that every
    stored value of the device format equals the caller's value at its recorded position
    (what cfs_hip_sym_update_values_* relies on), whole matrix and mirrored shards"""
    n, rp, ci, va, low = synth.generate(name, scale)
    rep = cfs.plan_check(n, rp, ci, va, options=cfs.make_options(flags=flags | 2048))
    assert rep["mismatches"] == 0 and rep["decoded"] == low
    rs = cfs.balanced_splits(n, rp, ci, 3)
    for r in range(3):
        rep = cfs.plan_check(n, rp, ci, va, 3, r, rs, options=cfs.make_options(flags=flags | 2048))
        assert rep["mismatches"] == 0


def test_hyb_takes_the_single_use_halo_columns_out():
    """ldoor stand-in: 2 % fat rows scatter single entries over a 40 000-row window;
    each of them costs a halo slot (slot table, x gather, strip, fold) for one nonzero"""
    n, rp, ci, va, low = synth.generate("ldoor", 0.3)
    plain = cfs.plan_check(n, rp, ci, va)
    hyb = cfs.plan_check(n, rp, ci, va, options=cfs.make_options(flags=FLAG_HYB))
    assert hyb["mismatches"] == 0
    assert hyb["far_entries"] > 0.02 * low
    assert hyb["halo_slots"] < 0.7 * plain["halo_slots"]


@pytest.mark.parametrize("nranks", [2, 3])
@pytest.mark.parametrize("flags", [FLAG_HYB, FLAG_HYB | FLAG_EXCHANGE])
def test_hyb_shards(nranks, flags):
    n, rp, ci, va, low = synth.generate("ldoor", 0.1)
    rs = cfs.balanced_splits(n, rp, ci, nranks)
    tot = 0
    for r in range(nranks):
        rep = cfs.plan_check(n, rp, ci, va, nranks, r, rs, options=cfs.make_options(flags=flags))
        assert rep["mismatches"] == 0
        tot += rep["nnz_low"]
    assert tot == low


@pytest.mark.parametrize("slots,block", [(64, 256), (128, 512), (777, 256), (2560, 512),
                                         (5120, 1024), (10240, 1024)])
@pytest.mark.parametrize("flags", [0, 8])
def test_schedule_options(slots, block, flags):
    n, rp, ci, va, low = synth.generate("pwtk", 0.05)
    rep = cfs.plan_check(n, rp, ci, va, options=cfs.make_options(slots, 0, block, flags))
    assert rep["mismatches"] == 0 and rep["decoded"] == low
    assert rep["lds_slots"] <= max(64, (min(slots, 10 * block) + 63) // 64 * 64)


@pytest.mark.parametrize("n,avg,band", [(1, 0, None), (2, 1, None), (63, 2, None), (64, 5, None),
                                        (65, 5, None), (500, 30, 40), (3000, 4, None)])
def test_schedule_small_and_ragged(n, avg, band):
    n, rp, ci, va = synth.random_symmetric(n, avg, seed=n, band=band)
    rep = cfs.plan_check(n, rp, ci, va, options=cfs.make_options(max_slots=256))
    assert rep["mismatches"] == 0


def test_schedule_rows_without_diagonal_and_empty_rows():
    import scipy.sparse as sp
    n = 300
    rng = np.random.default_rng(0)
    L = sp.random(n, n, density=0.02, random_state=1, format="csr")
    L = sp.tril(L, k=-1).tolil()
    L[100:140, :] = 0  # rows with no lower entries
    A = (L + L.T + sp.diags(np.where(np.arange(n) % 7 == 0, 0.0, 1.0))).tocsr()
    A.eliminate_zeros()
    A.sort_indices()
    rep = cfs.plan_check(n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data)
    assert rep["mismatches"] == 0


def test_dense_row_is_rejected_not_mangled():
    import scipy.sparse as sp
    n = 400
    L = sp.lil_matrix((n, n))
    L[n - 1, :n - 1] = 1.0  # arrow: last row touches every column
    A = (L + L.T + sp.identity(n)).tocsr()
    A.sort_indices()
    rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    with pytest.raises(_lib.CfsHipError, match="dense row"):
        cfs.plan_check(n, rp, ci, A.data, options=cfs.make_options(max_slots=128, flags=8))
    # with clustering (the default) the hub row is numbered early and its entries
    # are stored at their other ends: the arrow becomes schedulable
    rep = cfs.plan_check(n, rp, ci, A.data, options=cfs.make_options(max_slots=128))
    assert rep["mismatches"] == 0 and rep["decoded"] == n - 1


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_shard_schedules(nranks):
    n, rp, ci, va, low = synth.generate("Flan_1565", 0.02)
    rs = cfs.balanced_splits(n, rp, ci, nranks)
    assert rs[0] == 0 and rs[-1] == n and np.all(np.diff(rs) >= 0)
    assert all(int(v) % 16 == 0 for v in rs[1:-1])  # BlkFactor alignment, csr_matrix.tpp:418
    tot, sent, mirrored, offblock_low = 0, 0, 0, 0
    xopt = cfs.make_options(flags=cfs.FLAG_SHARD_EXCHANGE)
    for r in range(nranks):
        # default form: off-block entries mirrored, nothing to send
        rep = cfs.plan_check(n, rp, ci, va, nranks, r, rs)
        assert rep["mismatches"] == 0 and rep["remote_vals"] == 0
        mirrored += rep["mirror_entries"]
        lo = np.repeat(np.arange(n), np.diff(rp))
        offblock_low += int(np.sum((lo >= rs[r]) & (lo < rs[r + 1]) & (ci < rs[r])))
        # exchange form: contributions to lower ranks are packed
        rep = cfs.plan_check(n, rp, ci, va, nranks, r, rs, options=xopt)
        assert rep["mismatches"] == 0 and rep["mirror_entries"] == 0
        tot += rep["nnz_low"]
        counts, rows = cfs.plan_send_info(n, rp, ci, va, nranks, r, rs)
        assert counts.sum() == rows.size == rep["remote_vals"]
        assert np.all(counts[r:] == 0)  # contributions only flow to LOWER ranks
        assert np.all(rows < rs[r]) and np.all(np.diff(rows) > 0)
        sent += rows.size
    assert tot == low
    # every off-block lower entry is stored a second time by the rank that owns its column
    assert mirrored == offblock_low > 0
    # nnz balance within 10 %
    per = [cfs.plan_check(n, rp, ci, va, nranks, r, rs)["nnz_low"] for r in range(nranks)]
    assert max(per) <= 1.1 * (low / nranks) + 1000


def test_duplicates_and_unsymmetric_structure_keep_natural_order():
    """the clustered order pairs every entry with its mirror image; input with
    duplicate entries or a missing mirror entry must not be mangled by it"""
    import scipy.sparse as sp
    n = 600
    rng = np.random.default_rng(5)
    L = sp.random(n, n, density=0.02, random_state=2, format="coo")
    L = sp.tril(L, k=-1).tocoo()
    rows = np.concatenate([L.row, L.col, np.arange(n)])
    cols = np.concatenate([L.col, L.row, np.arange(n)])
    vals = np.concatenate([L.data, L.data, np.full(n, 3.0)])
    # duplicate one lower entry (and its mirror) with a different value, like the
    # Matrix-Market reader does for repeated lines
    r0, c0 = int(L.row[0]), int(L.col[0])
    rows = np.concatenate([rows, [r0, c0]])
    cols = np.concatenate([cols, [c0, r0]])
    vals = np.concatenate([vals, [0.123, 0.123]])
    order = np.lexsort((np.arange(rows.size), cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    rp = np.zeros(n + 1, dtype=np.int32)
    np.add.at(rp, rows + 1, 1)
    rp = np.cumsum(rp).astype(np.int32)
    rep = cfs.plan_check(n, rp, cols.astype(np.int32), vals, options=cfs.make_options(flags=16))
    assert rep["mismatches"] == 0 and rep["decoded"] == L.nnz + 1
    # structurally unsymmetric: drop one upper entry
    keep = ~((rows == c0) & (cols == r0))
    rp2 = np.zeros(n + 1, dtype=np.int32)
    np.add.at(rp2, rows[keep] + 1, 1)
    rp2 = np.cumsum(rp2).astype(np.int32)
    rep = cfs.plan_check(n, rp2, cols[keep].astype(np.int32), vals[keep],
                         options=cfs.make_options(flags=16))
    assert rep["mismatches"] == 0
    # no duplicates at all, only missing upper images (every 7th): an entry whose
    # column comes later in the clustered schedule than its row would be lost
    A = (L + L.T + sp.diags(np.full(n, 3.0))).tocsr()
    A.sort_indices()
    ar = np.repeat(np.arange(n), np.diff(A.indptr))
    up = np.flatnonzero(A.indices > ar)[::7]
    A.data[up] = 0
    A.eliminate_zeros()
    rep = cfs.plan_check(n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data,
                         options=cfs.make_options(flags=16))
    assert rep["mismatches"] == 0 and rep["decoded"] == L.nnz


def test_mirrored_shard_refuses_unsymmetric_offblock_structure():
    """a mirrored shard finds the entries of higher ranks' rows through their images in
    its own rows: a missing image (either way) must be an error, never a silent loss;
    the exchange form reads the lower triangle only and takes the matrix"""
    import scipy.sparse as sp
    n, rp, ci, va, _ = synth.generate("pwtk", 0.02)
    rs = cfs.balanced_splits(n, rp, ci, 2)
    A = sp.csr_matrix((va, ci, rp), shape=(n, n)).tolil()
    cut = int(rs[1])
    rows = np.repeat(np.arange(n), np.diff(rp))
    k = int(np.flatnonzero((rows >= cut) & (ci < cut))[0])   # a lower entry across the cut
    r, c = int(rows[k]), int(ci[k])
    for drop in ((c, r), (r, c)):                              # its image / the entry itself
        B = A.copy()
        B[drop[0], drop[1]] = 0
        B = B.tocsr()
        B.eliminate_zeros()
        B.sort_indices()
        args = (n, B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data)
        # the rank that owns the column side of the cut entry notices; dist.build_shard
        # makes all ranks agree and fall back to the exchange form
        with pytest.raises(_lib.CfsHipError, match="mirror"):
            cfs.plan_check(*args, 2, 0, rs)
        assert cfs.plan_check(*args, 2, 1, rs)["mismatches"] == 0
        for rank in (0, 1):
            rep = cfs.plan_check(*args, 2, rank, rs,
                                 options=cfs.make_options(flags=cfs.FLAG_SHARD_EXCHANGE))
            assert rep["mismatches"] == 0
