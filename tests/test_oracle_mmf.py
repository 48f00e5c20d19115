"""Oracle, part 1: Matrix-Market reader + CSR constructor
(include/io/mmf.hpp:179-343, src/mmf.cpp:6-44, csr_matrix.tpp:8-111).

Pinned against the GENUINE reference reader: tests/golden/*.ref.npz hold what
oracle/_ref/ref_mmf_dump (the reference's own src/mmf.cpp + io/mmf.hpp, compiled
where they lie) returned for the committed .mtx inputs; when the binary is
present (build container) the comparison is also run live."""
import glob
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_mmf_dump")
FILES = sorted(glob.glob(os.path.join(GOLD, "*.mtx")))


def csr_to_coo1(m):
    """CSR (0-based) -> one-based (row, col, val) stream in stored order"""
    rows = np.repeat(np.arange(m["nrows"]), np.diff(m["rowptr"])) + 1
    return rows.astype(np.int32), (m["colind"] + 1).astype(np.int32), m["values"]


def canon(r, c, v):
    """order-insensitive among duplicates (std::sort leaves it unspecified)"""
    o = np.lexsort((v, c, r))
    return r[o], c[o], v[o]


def test_fixtures_present():
    assert len(FILES) >= 8


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
@pytest.mark.parametrize("dtype,key", [(np.float64, "val64"), (np.float32, "val32")])
def test_reader_matches_reference_golden(path, dtype, key):
    g = np.load(path[:-4] + ".ref.npz")
    m = oracle.mmf_load(path, dtype)
    assert [m["nrows"], m["ncols"], m["nnz"], int(m["symmetric"])] == list(g["header"])
    r, c, v = csr_to_coo1(m)
    gr, gc, gv = g["rowcol"][:, 0], g["rowcol"][:, 1], g[key]
    # (row, col) stream is bit-exact and sorted exactly as the reference sorts
    assert np.array_equal(r, gr) and np.array_equal(c, gc)
    a, b = canon(r, c, v), canon(gr, gc, gv)
    assert np.array_equal(a[2].view(np.uint8), b[2].view(np.uint8))  # bit-exact values
    # CSR invariants of csr_matrix.tpp:74-107
    assert m["rowptr"][0] == 0 and m["rowptr"][-1] == m["nnz"]
    assert np.all(np.diff(m["rowptr"]) >= 0)


@pytest.mark.skipif(not os.path.exists(DUMP), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_reader_matches_reference_live(path, tmp_path):
    out = str(tmp_path / "o.bin")
    subprocess.run([DUMP, path, "f64", out], check=True, stdout=subprocess.DEVNULL)
    raw = open(out, "rb").read()
    hdr = np.frombuffer(raw, np.int32, 4)
    nnz = int(hdr[2])
    rc = np.frombuffer(raw, np.int32, 2 * nnz, 16).reshape(nnz, 2)
    val = np.frombuffer(raw, np.float64, nnz, 16 + 8 * nnz)
    m = oracle.mmf_load(path, np.float64)
    r, c, v = csr_to_coo1(m)
    assert np.array_equal(r, rc[:, 0]) and np.array_equal(c, rc[:, 1])
    assert np.array_equal(canon(r, c, v)[2], canon(rc[:, 0], rc[:, 1], val)[2])


@pytest.mark.skipif(not os.path.exists(DUMP), reason="oracle/_ref not built (no /root/reference)")
def test_reader_live_on_synthetic_stand_in(tmp_path):
    """a bigger structured case: the pdb1HYS stand-in written by our generator"""
    from cfs_spmv_amd import synth
    n, rp, ci, va, _ = synth.generate("pdb1HYS", 0.05)
    p = str(tmp_path / "m.mtx")
    synth.write_mtx(p, n, rp, ci, va)
    out = str(tmp_path / "o.bin")
    subprocess.run([DUMP, p, "f64", out], check=True, stdout=subprocess.DEVNULL)
    raw = open(out, "rb").read()
    hdr = np.frombuffer(raw, np.int32, 4)
    nnz = int(hdr[2])
    rc = np.frombuffer(raw, np.int32, 2 * nnz, 16).reshape(nnz, 2)
    val = np.frombuffer(raw, np.float64, nnz, 16 + 8 * nnz)
    m = oracle.mmf_load(p, np.float64)
    r, c, v = csr_to_coo1(m)
    assert np.array_equal(r, rc[:, 0]) and np.array_equal(c, rc[:, 1]) and np.array_equal(v, val)
    # and the file round-trips to the generator's CSR: the bit-exact row_ptr/col_idx contract
    assert np.array_equal(m["rowptr"], rp) and np.array_equal(m["colind"], ci)
    assert np.array_equal(m["values"], va)


def _w(tmp_path, text, name="t.mtx"):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


def test_symmetric_expansion_and_sort(tmp_path):
    p = _w(tmp_path, "%%MatrixMarket matrix coordinate real symmetric\n3 3 4\n"
                     "3 1 5.0\n1 1 1.0\n2 2 2.0\n3 3 3.0\n")
    m = oracle.mmf_load(p)
    assert m["symmetric"] and m["nnz"] == 5  # off-diagonal mirrored (mmf.hpp:279-293)
    assert list(m["rowptr"]) == [0, 2, 3, 5]
    assert list(m["colind"]) == [0, 2, 1, 0, 2]
    assert list(m["values"]) == [1.0, 5.0, 2.0, 5.0, 3.0]


def test_pattern_value_and_header_tokens(tmp_path):
    p = _w(tmp_path, "%%MatrixMarket matrix coordinate pattern general\n2 2 2\n1 2\n2 2\n")
    m = oracle.mmf_load(p)
    assert list(m["values"]) == [0.42, 0.42]  # mmf.hpp:334-337
    assert not m["symmetric"]
    p = _w(tmp_path, "%%MatrixMarket matrix coordinate real general base-0\n2 2 2\n0 0 1\n1 1 2\n")
    m = oracle.mmf_load(p)
    assert list(m["colind"]) == [0, 1] and list(m["rowptr"]) == [0, 1, 2]


def test_explicit_zeros_kept_empty_rows_repeat(tmp_path):
    p = _w(tmp_path, "%%MatrixMarket matrix coordinate real general\n4 4 3\n1 1 0.0\n3 2 7\n4 4 1\n")
    m = oracle.mmf_load(p)
    assert m["nnz"] == 3 and list(m["values"]) == [0.0, 7.0, 1.0]
    assert list(m["rowptr"]) == [0, 1, 1, 2, 3]  # csr_matrix.tpp:91-96


@pytest.mark.parametrize("text,code", [
    ("%%MatrixMarket matrix array real general\n2 2 1\n1 1 1\n", -5),         # not coordinate
    ("%%MatrixMarket matrix coordinate real skew-symmetric\n2 2 1\n2 1 1\n", -6),
    ("%%MatrixMarket matrix coordinate real\n2 2 1\n1 1 1\n", -4),            # < 5 header tokens
    ("%%Bogus matrix coordinate real general\n2 2 1\n1 1 1\n", -3),
    ("%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1\n", -9),    # file ends early
    ("%%MatrixMarket matrix coordinate real general\n2 2 1\n1 1 1", -9),      # no trailing newline
    ("%%MatrixMarket matrix coordinate real general\n2 2 1\n1\t1\t1\n", -8),  # tabs do not split
    ("%%MatrixMarket matrix coordinate real general\n3 3 1\n1 1 1\n", -11),   # trailing empty rows
])
def test_malformed_inputs_are_rejected(tmp_path, text, code):
    """where the reference prints and exit(1)s / asserts, the oracle returns a code"""
    p = _w(tmp_path, text)
    with pytest.raises(ValueError) as e:
        oracle.mmf_load(p)
    assert e.value.args[0] == code


def test_missing_file(tmp_path):
    with pytest.raises(ValueError) as e:
        oracle.mmf_load(str(tmp_path / "nope.mtx"))
    assert e.value.args[0] == -1


def test_fp32_values_round_once(tmp_path):
    p = _w(tmp_path, "%%MatrixMarket matrix coordinate real general\n1 1 1\n1 1 0.1\n")
    m = oracle.mmf_load(p, np.float32)
    assert m["values"].dtype == np.float32 and m["values"][0] == np.float32(0.1)
