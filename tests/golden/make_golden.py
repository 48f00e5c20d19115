#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the BUILD container only).

  *.mtx            small Matrix-Market inputs written by this script (our data)
  *.ref.npz        what the GENUINE reference reader (oracle/_ref/ref_mmf_dump,
                   i.e. /root/reference/src/mmf.cpp + include/io/mmf.hpp compiled
                   where they lie) returned for that file: nrows, ncols, nnz,
                   symmetric, and the sorted one-based (row, col, val) stream,
                   for fp64 and fp32
  *.exact.npz      x (seeded) and y = A x computed in exact rational arithmetic
                   (python fractions) and rounded once to fp64 -- independent
                   of both the oracle and the HIP path

The reference's numeric path (csr_matrix.tpp) is unbuildable in this image
(needs TBB headers + an autoheader config.h), so no reference y vectors exist;
see DESIGN.md "Oracle".  Fixtures are data only: no reference source text.
"""
import os
import subprocess
import sys
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_mmf_dump")


def write(path, text):
    with open(path, "w") as f:
        f.write(text)


def sym_lines(n, entries):
    return "".join(f"{r} {c} {v!r}\n" for r, c, v in entries)


def make_files():
    rng = np.random.default_rng(20261004)
    files = {}
    # g01: small symmetric, full diagonal, comment lines before the size line
    n = 12
    ent = [(i, i, float(np.round(rng.uniform(1, 3), 6))) for i in range(1, n + 1)]
    for i in range(2, n + 1):
        for j in range(1, i):
            if rng.random() < 0.3:
                ent.append((i, j, float(np.round(rng.uniform(-1, 1), 6))))
    rng.shuffle(ent)
    files["g01_sym_small.mtx"] = ("%%MatrixMarket matrix coordinate real symmetric\n"
                                  "% a comment\n%another\n"
                                  f"{n} {n} {len(ent)}\n" + sym_lines(n, ent))
    # g02: banded symmetric n=200, ~6 lower per row
    n = 200
    ent = [(i, i, float(np.round(rng.uniform(2, 4), 8))) for i in range(1, n + 1)]
    for i in range(2, n + 1):
        for j in set(int(v) for v in rng.integers(max(1, i - 30), i, size=6)):
            ent.append((i, j, float(np.round(rng.uniform(-1, 1), 8))))
    files["g02_sym_banded.mtx"] = ("%%MatrixMarket matrix coordinate real symmetric\n"
                                   f"{n} {n} {len(ent)}\n" + sym_lines(n, ent))
    # g03: pattern symmetric (2-token lines -> 0.42), extra blanks between tokens
    n = 30
    ent = [(i, i) for i in range(1, n + 1)]
    for i in range(2, n + 1):
        for j in set(int(v) for v in rng.integers(1, i, size=3)):
            ent.append((i, j))
    files["g03_sym_pattern.mtx"] = ("%%MatrixMarket matrix coordinate pattern symmetric\n"
                                    f"{n} {n} {len(ent)}\n"
                                    + "".join(f"  {r}   {c} \n" for r, c in ent))
    # g04: general (unsymmetric), unsorted input, integer field
    n, m = 25, 25
    ent = []
    for i in range(1, n + 1):
        ent.append((i, i, 2))
        for j in set(int(v) for v in rng.integers(1, m + 1, size=3)):
            if j != i:
                ent.append((i, j, int(rng.integers(-5, 6))))
    rng.shuffle(ent)
    files["g04_general.mtx"] = ("%%MatrixMarket matrix coordinate integer general\n"
                                f"{n} {m} {len(ent)}\n"
                                + "".join(f"{r} {c} {v}\n" for r, c, v in ent))
    # g05: zero-based symmetric with the base-0 header token
    n = 16
    ent = [(i, i, 1.5 + i) for i in range(n)]
    for i in range(1, n):
        ent.append((i, i - 1, -0.25 * i))
    files["g05_sym_base0.mtx"] = ("%%MatrixMarket matrix coordinate real symmetric base-0\n"
                                  f"{n} {n} {len(ent)}\n" + sym_lines(n, ent))
    # g06: no banner (file_mode 1): first line is the size line; general
    n = 10
    ent = [(i, i, float(i)) for i in range(1, n + 1)] + [(1, n, 0.5), (n, 1, -0.5)]
    files["g06_nobanner.mtx"] = f"{n} {n} {len(ent)}\n" + sym_lines(n, ent)
    # g07: explicit zeros are kept, duplicates are kept (symmetric)
    n = 8
    ent = [(i, i, 2.0) for i in range(1, n + 1)] + [(3, 1, 0.0), (5, 2, 0.75), (5, 2, 0.25),
                                                    (8, 7, -1.0)]
    files["g07_sym_zeros_dups.mtx"] = ("%%MatrixMarket matrix coordinate real symmetric\n"
                                       f"{n} {n} {len(ent)}\n" + sym_lines(n, ent))
    # g08: stencil-like symmetric n=343 (7^3 grid, 7-point) -- structured case
    g = 7
    n = g ** 3
    ent = []
    for z in range(g):
        for y in range(g):
            for x in range(g):
                i = x + g * (y + g * z) + 1
                ent.append((i, i, 6.5))
                if x:
                    ent.append((i, i - 1, -1.0))
                if y:
                    ent.append((i, i - g, -1.0 - 0.125 * x))
                if z:
                    ent.append((i, i - g * g, -0.5))
    files["g08_sym_stencil.mtx"] = ("%%MatrixMarket matrix coordinate real symmetric\n"
                                    f"{n} {n} {len(ent)}\n" + sym_lines(n, ent))
    return files


def ref_dump(path, kind):
    out = path + f".{kind}.bin"
    subprocess.run([DUMP, path, kind, out], check=True, stdout=subprocess.DEVNULL)
    raw = open(out, "rb").read()
    os.remove(out)
    hdr = np.frombuffer(raw, dtype=np.int32, count=4)
    nnz = int(hdr[2])
    rc = np.frombuffer(raw, dtype=np.int32, count=2 * nnz, offset=16).reshape(nnz, 2)
    vt = np.float64 if kind == "f64" else np.float32
    val = np.frombuffer(raw, dtype=vt, count=nnz, offset=16 + 8 * nnz)
    return hdr.copy(), rc.copy(), val.copy()


def exact_y(nrows, ncols, rc, val, x):
    y = [Fraction(0)] * nrows
    for (r, c), v in zip(rc, val):
        y[r - 1] += Fraction(float(v)) * Fraction(float(x[c - 1]))
    ya = [Fraction(0)] * nrows
    for (r, c), v in zip(rc, val):
        ya[r - 1] += abs(Fraction(float(v)) * Fraction(float(x[c - 1])))
    return np.array([float(v) for v in y]), np.array([float(v) for v in ya])


def main():
    if not os.path.exists(DUMP):
        sys.exit("oracle/_ref/ref_mmf_dump missing: run `make -C oracle` with /root/reference present")
    for name, text in make_files().items():
        path = os.path.join(HERE, name)
        write(path, text)
        h64, rc64, v64 = ref_dump(path, "f64")
        h32, rc32, v32 = ref_dump(path, "f32")
        assert (rc64 == rc32).all()
        np.savez_compressed(path[:-4] + ".ref.npz", header=h64, rowcol=rc64, val64=v64, val32=v32)
        rng = np.random.default_rng(len(name) * 7919)
        x = np.round(rng.uniform(-1.0, 1.0, size=int(h64[1])), 10)
        y, ya = exact_y(int(h64[0]), int(h64[1]), rc64, v64, x)
        np.savez_compressed(path[:-4] + ".exact.npz", x=x, y=y, absrow=ya)
        print(f"{name}: nrows={h64[0]} ncols={h64[1]} nnz={h64[2]} symmetric={h64[3]}")


if __name__ == "__main__":
    main()
