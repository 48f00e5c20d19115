"""Randomised structural check of the tile schedule (CPU, no GPU): for matrices of
many shapes -- banded, random, with fat rows, hub rows, empty rows, missing
diagonals, repeated node-like row blocks -- and random schedule options and shard
counts, the device format must decode back to exactly the stored entries
(cfs_hip_sym_plan_check: multiset of (row, col, value bits), slot classes, fold and
send coverage, tiles partition the rows)."""
import numpy as np
import pytest
import scipy.sparse as sp

import cfs_spmv_amd as cfs
from cfs_spmv_amd import _lib
from rand_matrices import random_matrix


@pytest.mark.parametrize("seed", range(240))
def test_random_schedules_decode_to_the_input(seed):
    rng = np.random.default_rng(1000 + seed)
    kind = ["band", "random", "nodes", "hub"][seed % 4]
    n = int(rng.integers(300, 4000))
    n, A = random_matrix(rng, n, kind)
    rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    va = A.data.astype(np.float64 if seed % 3 else np.float32)
    low = int(sp.tril(A, -1).nnz)
    block = int(rng.choice([256, 512, 1024]))
    slots = int(rng.choice([0, 256, 1024, 2560]))
    flags = int(rng.choice([0, 0, 8, 16]))
    nranks = int(rng.choice([1, 1, 2, 3, 5]))
    if nranks > 1 and rng.random() < 0.4:
        flags |= cfs.FLAG_SHARD_EXCHANGE
    opt = cfs.make_options(max_slots=slots, block_threads=block, flags=flags)
    rs = cfs.balanced_splits(n, rp, ci, nranks) if nranks > 1 else None
    tot = 0
    for rank in range(nranks):
        try:
            rep = cfs.plan_check(n, rp, ci, va, nranks, rank, rs, options=opt)
        except _lib.CfsHipError as e:
            # the only legitimate refusal: a row that does not fit the window
            assert "dense row" in str(e), str(e)
            assert kind == "hub" or slots == 256
            return
        assert rep["mismatches"] == 0, (kind, n, block, slots, flags, nranks, rank, rep)
        assert rep["decoded"] == rep["nnz_low"] + rep["mirror_entries"]
        tot += rep["nnz_low"]
    assert tot == low
