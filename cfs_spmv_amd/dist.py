"""1-D row-block sharding of the symmetric SpMV over torch.distributed
(one process per GPU; backend "nccl" is RCCL over xGMI on ROCm).

SURVEY.md 8(e): rank g owns rows [R_g, R_{g+1}) (boundaries at multiples of 16,
nnz_low-balanced), holds their strict-lower entries + diagonal and a full
replica of x.  Row-side sums are local.  Transposed updates y_j += a_ij x_i with
j < R_g are the reference's *direct conflicts* (csr_matrix.tpp:1443-1451); they
are packed (one value per touched remote row) and exchanged in ONE collective.

The north-star names a reduce-scatter of the off-block contributions.  A dense
reduce-scatter would move n*s bytes per rank around a ring (~72 us for
Flan_1565 over xGMI); the contributions only reach the previous
block(s), so the exchange here is the sparse form of the same reduction: an
all-to-all of exactly the touched rows, summed on the owner in a fixed order.
"""
import numpy as np


class ShardedSym:
    """backend: object with send_counts(), send_rows(), set_recv(rows),
    spmv_local(y_block, x, send), recv_fold(y_block, recv), row_begin, row_end.
    On a GPU box that is cfs_spmv_amd.SymMatrix; CPU tests pass a double."""

    def __init__(self, backend, nranks, rank, dtype, device, pg=None, stage_via_host=False):
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
        tile kernel -> pack -> [exchange || local fold] -> fold of what arrived"""
        self.A.spmv_phases(y_block, x, self.send_buf, 1 | 4)   # tiles + pack
        self.finish(y_block, x)

    def finish(self, y_block, x):
        """everything after tiles + pack: start the one collective of the path,
        fold the local strips while it is in flight, then fold what arrived"""
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
