"""1-D row-block sharding of the symmetric SpMV over torch.distributed
(one process per GPU; backend "nccl" is RCCL over xGMI on ROCm).

SURVEY.md 8(e): rank g owns rows [R_g, R_{g+1}) (boundaries at multiples of 16,
nnz_low-balanced), holds their strict-lower entries + diagonal and a full
replica of x.  Row-side sums are local.  Transposed updates y_j += a_ij x_i with
j < R_g are the reference's *direct conflicts* (csr_matrix.tpp:1443-1451); they
are packed (one value per touched remote row) and exchanged in ONE collective.

Three forms, selected by `exchange`:

"none" (default)  -- MIRRORED shards: an off-block entry a_ij (i on rank g, j on a
    lower rank) is stored by both ranks and processed one-sided by each, so no
    contribution to y ever leaves a rank and an SpMV is one local launch sequence
    with NO collective (x is replicated anyway; the duplicated boundary entries
    are a few per cent of a shard).  A sharded SpMV of a ~0.1 ms matrix is bound
    by latency, and the cheapest exchange is the one that does not happen.
"all_to_all"      -- the exchange form: contributions to rows of lower ranks are
    packed (one value per touched remote row) and exchanged in ONE collective,
    summed on the owner in a fixed order: the sparse form of the reduce-scatter
    the north-star names (the contributions only reach the previous block(s)).
"reduce_scatter"  -- the dense form of the same: ONE reduce_scatter_tensor(sum)
    over nranks equal padded blocks; moves n*s bytes per rank around the ring
    (~72 us for Flan_1565 over xGMI).
"""
import numpy as np


class ShardedSym:
    """backend: object with send_counts(), send_rows(), set_recv(rows),
    spmv_local(y_block, x, send), recv_fold(y_block, recv), row_begin, row_end.
    On a GPU box that is cfs_spmv_amd.SymMatrix; CPU tests pass a double."""

    def __init__(self, backend, nranks, rank, dtype, device, pg=None, stage_via_host=False,
                 exchange="all_to_all", row_splits=None):
        # (the class default stays the exchange form: a backend double in the CPU
        # tests has no mirrored entries; build_shard() below defaults to "none")
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.A, self.nranks, self.rank, self.pg = backend, nranks, rank, pg
        self.device = device
        # rehearsal mode (several ranks on ONE GPU over gloo): collectives run on
        # host copies of the device buffers.  Never used with the nccl backend.
        self.stage = bool(stage_via_host)
        cdev = torch.device("cpu") if self.stage else device
        tdt = torch.float64 if np.dtype(dtype) == np.float64 else torch.float32
        send_counts = np.asarray(backend.send_counts(), dtype=np.int64)
        send_rows = np.asarray(backend.send_rows(), dtype=np.int32)
        assert send_counts.sum() == send_rows.size
        self.exchange = exchange
        if exchange == "none":
            # mirrored shard: nothing is sent, nothing arrives, no collective
            if send_rows.size:
                raise ValueError("exchange='none' needs a mirrored shard (the handle was built "
                                 "with CFS_HIP_FLAG_SHARD_EXCHANGE)")
            self.nsend = self.nrecv = 0
            self.send_buf = torch.zeros(1, dtype=tdt, device=device)
            self.recv_buf = self.send_buf
            return
        # 1) counts
        sc = torch.from_numpy(send_counts).to(cdev)
        rc = torch.zeros(nranks, dtype=torch.int64, device=cdev)
        dist.all_to_all_single(rc, sc, group=pg)
        self.send_splits = [int(v) for v in send_counts]
        self.recv_splits = [int(v) for v in rc.cpu().numpy()]
        # 2) row lists (static): what will arrive where
        nrecv = sum(self.recv_splits)
        sr = torch.from_numpy(send_rows.astype(np.int32)).to(cdev)
        rr = torch.zeros(nrecv, dtype=torch.int32, device=cdev)
        dist.all_to_all_single(rr, sr, self.recv_splits, self.send_splits, group=pg)
        backend.set_recv(rr.cpu().numpy())
        self.send_buf = torch.zeros(max(1, send_rows.size), dtype=tdt, device=device)
        self.recv_buf = torch.zeros(max(1, nrecv), dtype=tdt, device=device)
        self.nsend, self.nrecv = int(send_rows.size), int(nrecv)
        # exchange = "reduce_scatter": the dense form the north-star names -- every
        # rank scatters its packed contributions into a zero vector of nranks
        # equal (padded) blocks and ONE reduce-scatter(sum) hands each owner the sum
        # for its block.  Moves nranks*max_rows values per rank instead of the few
        # touched rows, so it is the slower option; kept selectable.
        if exchange == "reduce_scatter":
            if row_splits is None:
                raise ValueError("reduce_scatter exchange needs row_splits")
            rs = np.asarray(row_splits, dtype=np.int64)
            self.rs_max_rows = int(np.max(np.diff(rs)))
            owner = np.searchsorted(rs, send_rows, side="right") - 1
            pos = owner * self.rs_max_rows + (send_rows - rs[owner])
            self.rs_pos = torch.from_numpy(pos.astype(np.int64)).to(device)
            self.rs_dense = torch.zeros(self.rs_max_rows * nranks, dtype=tdt, device=device)
            self.rs_out = torch.zeros(self.rs_max_rows, dtype=tdt, device=device)
        elif exchange != "all_to_all":
            raise ValueError(f"unknown exchange {exchange!r}")

    def setup_allgather(self, row_splits):
        """prepare y -> x all-gather (iterative solvers feed the product back as the
        next input vector): equal-size padded blocks, one
        all_gather_into_tensor"""
        self.row_splits = [int(v) for v in row_splits]
        self.max_rows = max(b - a for a, b in zip(self.row_splits[:-1], self.row_splits[1:]))
        dt = self.send_buf.dtype
        dev = self.torch.device("cpu") if self.stage else self.device
        self._ag_in = self.torch.zeros(self.max_rows, dtype=dt, device=dev)
        self._ag_out = self.torch.zeros(self.max_rows * self.nranks, dtype=dt, device=dev)

    def allgather_rows(self, y_block, x_full):
        """x_full[row_splits[r]:row_splits[r+1]] <- rank r's y_block, for every r"""
        rows = y_block.numel()
        self._ag_in[:rows].copy_(y_block)
        self.dist.all_gather_into_tensor(self._ag_out, self._ag_in, group=self.pg)
        for r in range(self.nranks):
            a, b = self.row_splits[r], self.row_splits[r + 1]
            x_full[a:b].copy_(self._ag_out[r * self.max_rows:r * self.max_rows + (b - a)])

    def spmv(self, y_block, x):
        """y_block <- rows [row_begin,row_end) of A x; x is the full vector.
        mirrored shard: tile kernel -> fold, no collective.  exchange forms:
        tile kernel -> pack -> [exchange || local fold] -> fold of what arrived"""
        if self.exchange == "none":
            self.A.spmv_phases(y_block, x, None, 1 | 2)
            return
        self.A.spmv_phases(y_block, x, self.send_buf, 1 | 4)   # tiles + pack
        self.finish(y_block, x)

    def finish(self, y_block, x):
        """everything after tiles + pack: start the one collective of the path,
        fold the local strips while it is in flight, then fold what arrived"""
        if self.exchange == "none":
            self.A.spmv_phases(y_block, x, None, 2)            # local fold only
            return
        if self.exchange == "reduce_scatter":
            return self._finish_reduce_scatter(y_block, x)
        if self.stage:
            sh = self.send_buf[:self.nsend].cpu()
            rh = self.torch.zeros(self.nrecv, dtype=sh.dtype)
            self.dist.all_to_all_single(rh, sh, self.recv_splits, self.send_splits, group=self.pg)
            self.recv_buf[:self.nrecv].copy_(rh)
            work = None
        else:
            work = self.dist.all_to_all_single(
                self.recv_buf[:self.nrecv], self.send_buf[:self.nsend], self.recv_splits,
                self.send_splits, group=self.pg, async_op=True)
        self.A.spmv_phases(y_block, x, self.send_buf, 2)       # local fold
        if work is not None:
            work.wait()                                        # current stream waits for RCCL
        self.A.recv_fold(y_block, self.recv_buf)

    def _finish_reduce_scatter(self, y_block, x):
        torch, dist = self.torch, self.dist
        self.rs_dense.zero_()
        if self.nsend:
            self.rs_dense.index_copy_(0, self.rs_pos, self.send_buf[:self.nsend])
        backend = dist.get_backend(self.pg)
        if backend == "gloo" or self.stage:  # CPU rehearsal: gloo has no reduce_scatter_tensor
            tmp = self.rs_dense.cpu()
            dist.all_reduce(tmp, group=self.pg)
            out = tmp[self.rank * self.rs_max_rows:(self.rank + 1) * self.rs_max_rows]
            self.rs_out.copy_(out)
            work = None
        else:
            work = dist.reduce_scatter_tensor(self.rs_out, self.rs_dense, group=self.pg,
                                              async_op=True)
        self.A.spmv_phases(y_block, x, self.send_buf, 2)       # local fold
        if work is not None:
            work.wait()
        y_block += self.rs_out[:y_block.numel()]


def build_shard(n, rowptr, colind, values, nranks, rank, row_splits, device, options=None,
                exchange="none", pg=None, stage_via_host=False):
    """Build this rank's row block and its ShardedSym.  exchange="none" asks for a
    mirrored shard; a matrix whose off-block structure cannot be mirrored
    (structurally unsymmetric or duplicate entries) makes SOME rank fail, so the
    ranks agree (one all-reduce at set-up) and all fall back to "all_to_all".
    Returns (SymMatrix, ShardedSym, exchange actually used)."""
    import torch
    import torch.distributed as dist
    from . import matrix as M
    from ._lib import CfsHipError, ERR_MIRROR, Options
    flags = options.flags if options is not None else 0
    base = (options.max_slots, options.max_tile_nnz, options.block_threads) if options is not None \
        else (0, 0, 0)
    dtype = np.asarray(values).dtype
    A = None
    if exchange == "none":
        ok = 1
        try:
            A = M.SymMatrix(n, rowptr, colind, values, options=Options(*base, flags & ~M.FLAG_SHARD_EXCHANGE),
                            row_splits=row_splits, rank=rank)
        except CfsHipError as e:
            if e.code != ERR_MIRROR:
                raise
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32,
                            device=torch.device("cpu") if stage_via_host else device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=pg)
        if int(flag.item()) == 0:
            if A is not None:
                A.close()
            A, exchange = None, "all_to_all"
    if A is None:
        A = M.SymMatrix(n, rowptr, colind, values, options=Options(*base, flags | M.FLAG_SHARD_EXCHANGE),
                        row_splits=row_splits, rank=rank)
    sh = ShardedSym(A, nranks, rank, dtype, device, pg=pg, stage_via_host=stage_via_host,
                    exchange=exchange, row_splits=row_splits)
    return A, sh, exchange
