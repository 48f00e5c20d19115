"""N > 1 path on CPU: world_size 2 and 3 over the `gloo` backend.

The exchange logic (cfs_spmv_amd/dist.py: counts + row lists all-to-all, the
per-step all-to-all of packed contributions, owner-side fold) is the code under
test.  The per-rank device work is replaced by a host DOUBLE that lives only in
this test (scipy arithmetic on the shard's rows) but takes its packing order --
send_counts / send_rows -- from the real host-side schedule
(cfs_hip_sym_plan_send_info), exactly what SymMatrix reports on a GPU box."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class HostShardDouble:
    def __init__(self, n, rp, ci, va, nranks, rank, rs):
        import scipy.sparse as sp
        import cfs_spmv_amd as cfs
        self.row_begin, self.row_end = int(rs[rank]), int(rs[rank + 1])
        A = sp.csr_matrix((va, ci, rp), shape=(n, n))
        blk = A[self.row_begin:self.row_end, :]
        self.L = sp.tril(blk, k=self.row_begin - 1).tocsr()  # strict lower of the block rows
        self.d = A.diagonal()[self.row_begin:self.row_end]
        self.counts, self.rows = cfs.plan_send_info(n, rp, ci, va, nranks, rank, rs)
        self.recv_rows = None

    def send_counts(self):
        return self.counts

    def send_rows(self):
        return self.rows

    def set_recv(self, rows):
        self.recv_rows = np.asarray(rows, dtype=np.int64)

    def spmv_phases(self, y_block, x, send, phases):
        # the double computes everything in the "tiles + pack" call; its local
        # fold (phase 2 alone) has nothing left to do
        if phases & 1:
            self.spmv_local(y_block, x, send)

    def spmv_local(self, y_block, x, send):
        xn = x.numpy()
        rb, re = self.row_begin, self.row_end
        yt = self.L.T @ xn[rb:re]            # transposed updates, all columns < re
        y = self.d * xn[rb:re] + self.L @ xn + yt[rb:re]
        y_block.numpy()[:] = y
        if self.rows.size:
            send.numpy()[:self.rows.size] = yt[self.rows]

    def recv_fold(self, y_block, recv):
        if self.recv_rows is not None and self.recv_rows.size:
            np.add.at(y_block.numpy(), self.recv_rows - self.row_begin,
                      recv.numpy()[:self.recv_rows.size])


def _worker(rank, world, port, name, scale, q, exchange="all_to_all"):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch
        import torch.distributed as dist
        import cfs_spmv_amd as cfs
        from cfs_spmv_amd import synth
        from cfs_spmv_amd.dist import ShardedSym
        from oracle import oracle
        dist.init_process_group("gloo", rank=rank, world_size=world)
        n, rp, ci, va, _ = synth.generate(name, scale)
        x = synth.make_x(n)
        rs = cfs.balanced_splits(n, rp, ci, world)
        be = HostShardDouble(n, rp, ci, va, world, rank, rs)
        rccl_branch = exchange == "reduce_scatter_rccl_branch"
        sh = ShardedSym(be, world, rank, np.float64, torch.device("cpu"),
                        exchange="reduce_scatter" if rccl_branch else exchange, row_splits=rs)
        calls = []
        if rccl_branch:
            # drive the NON-gloo branch of _finish_reduce_scatter (the one RCCL takes on
            # the GPU box): the process group claims to be "nccl" and
            # reduce_scatter_tensor(async_op=True) is served by a stand-in with the same
            # contract (sum over ranks, rank r receives block r, returns a Work)
            class Work:
                def wait(self):
                    calls.append("wait")

            class DistShim:
                def __getattr__(self, k):
                    return getattr(dist, k)

                def get_backend(self, pg=None):
                    return "nccl"

                def reduce_scatter_tensor(self, out, inp, group=None, async_op=False):
                    assert async_op and inp.numel() == out.numel() * world
                    tmp = inp.clone()
                    dist.all_reduce(tmp, group=group)
                    out.copy_(tmp[rank * out.numel():(rank + 1) * out.numel()])
                    calls.append("reduce_scatter_tensor")
                    return Work()
            sh.dist = DistShim()
        xt = torch.from_numpy(x.copy())
        yb = torch.full((be.row_end - be.row_begin,), 7.0, dtype=torch.float64)
        for _ in range(2):  # twice: buffers are reused
            sh.spmv(yb, xt)
        # y -> x all-gather (solver loop): every rank ends up with the whole product
        sh.setup_allgather(rs)
        full = torch.zeros(n, dtype=torch.float64)
        sh.allgather_rows(yb, full)
        y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
        den_all = np.maximum(np.abs(y_ld), absrow)
        assert float(np.max(np.abs(full.numpy() - y_ld) / den_all)) <= 1e-12
        sl = slice(be.row_begin, be.row_end)
        den = np.maximum(np.abs(y_ld[sl]), absrow[sl])
        err = float(np.max(np.abs(yb.numpy() - y_ld[sl]) / den)) if den.size else 0.0
        # what this rank received must be exactly the rows the others aimed at it
        ok_rows = bool(np.all((be.recv_rows >= be.row_begin) & (be.recv_rows < be.row_end))) \
            if be.recv_rows is not None and be.recv_rows.size else True
        if rccl_branch:  # the branch really ran: one collective + one wait per SpMV
            assert calls == ["reduce_scatter_tensor", "wait"] * 2, calls
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, err, ok_rows, int(sh.nsend), int(sh.nrecv)))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, f"{e}\n{traceback.format_exc()}", False, 0, 0))


@pytest.mark.parametrize("world,name,scale,exchange", [
    (2, "Flan_1565", 0.01, "all_to_all"), (3, "pwtk", 0.03, "all_to_all"),
    (2, "ldoor", 0.01, "all_to_all"), (3, "Flan_1565", 0.01, "reduce_scatter"),
    (2, "Flan_1565", 0.01, "reduce_scatter_rccl_branch")])
def test_sharded_exchange_gloo(world, name, scale, exchange):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, scale, q, exchange))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    res.sort()
    for rank, err, ok_rows, nsend, nrecv in res:
        assert not isinstance(err, str), f"rank {rank} failed: {err}"
        assert err <= 1e-12 and ok_rows, (rank, err)
    assert sum(r[3] for r in res) == sum(r[4] for r in res) > 0  # every packed value arrives
    assert res[0][3] == 0  # rank 0 owns the first rows: nothing to send


def _cg_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch
        import torch.distributed as dist
        import scipy.sparse as sp
        import scipy.sparse.linalg as spl
        import cfs_spmv_amd as cfs
        from cfs_spmv_amd import synth
        from cfs_spmv_amd.dist import ShardedSym
        from cfs_spmv_amd.solver import cg_sharded
        dist.init_process_group("gloo", rank=rank, world_size=world)
        n, rp, ci, va, _ = synth.generate("pwtk", 0.02)
        rs = cfs.balanced_splits(n, rp, ci, world)
        be = HostShardDouble(n, rp, ci, va, world, rank, rs)
        sh = ShardedSym(be, world, rank, np.float64, torch.device("cpu"))
        b = synth.make_x(n, 7)
        bb = torch.from_numpy(b[be.row_begin:be.row_end].copy())
        u, it, res = cg_sharded(sh, rs, bb, tol=1e-11, maxiter=400)
        u_ref = spl.spsolve(sp.csc_matrix(sp.csr_matrix((va, ci, rp), shape=(n, n))), b)
        err = float(np.max(np.abs(u.numpy() - u_ref[be.row_begin:be.row_end])) / np.max(np.abs(u_ref)))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, err, it, res))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, f"{e}\n{traceback.format_exc()}", 0, 0.0))


def test_sharded_cg_gloo():
    """solver-style loop over the sharded path: every product is fed back as the next
    input through the y -> x all-gather"""
    import torch.multiprocessing as mp
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cg_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, err, it, rr in res:
        assert not isinstance(err, str), f"rank {rank} failed: {err}"
        assert err <= 1e-9 and rr <= 1e-10 and 0 < it < 400, (rank, err, it, rr)
    assert res[0][2] == res[1][2]  # same iteration count on every rank


class HostMirroredDouble:
    """host stand-in of a MIRRORED shard: every entry that touches a row of the block is
    stored by this rank, so its block of y needs nothing from anyone"""

    def __init__(self, n, rp, ci, va, nranks, rank, rs):
        import scipy.sparse as sp
        self.row_begin, self.row_end = int(rs[rank]), int(rs[rank + 1])
        self.B = sp.csr_matrix((va, ci, rp), shape=(n, n))[self.row_begin:self.row_end, :]
        self.nranks = nranks

    def send_counts(self):
        return np.zeros(self.nranks, dtype=np.int32)

    def send_rows(self):
        return np.zeros(0, dtype=np.int32)

    def spmv_phases(self, y_block, x, send, phases):
        if phases & 1:
            y_block.numpy()[:] = self.B @ x.numpy()


def _mirror_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch
        import torch.distributed as dist
        import scipy.sparse as sp
        import scipy.sparse.linalg as spl
        import cfs_spmv_amd as cfs
        from cfs_spmv_amd import synth
        from cfs_spmv_amd.dist import ShardedSym
        from cfs_spmv_amd.solver import cg_sharded
        dist.init_process_group("gloo", rank=rank, world_size=world)
        n, rp, ci, va, _ = synth.generate("pwtk", 0.02)
        rs = cfs.balanced_splits(n, rp, ci, world)
        # the real schedule of this rank's mirrored shard: nothing to send
        rep = cfs.plan_check(n, rp, ci, va, world, rank, rs)
        assert rep["mismatches"] == 0 and rep["remote_vals"] == 0
        be = HostMirroredDouble(n, rp, ci, va, world, rank, rs)
        sh = ShardedSym(be, world, rank, np.float64, torch.device("cpu"), exchange="none")
        x = synth.make_x(n)
        yb = torch.full((be.row_end - be.row_begin,), 7.0, dtype=torch.float64)
        sh.spmv(yb, torch.from_numpy(x.copy()))
        A = sp.csr_matrix((va, ci, rp), shape=(n, n))
        err = float(np.max(np.abs(yb.numpy() - (A @ x)[be.row_begin:be.row_end])))
        b = synth.make_x(n, 7)
        u, it, res = cg_sharded(sh, rs, torch.from_numpy(b[be.row_begin:be.row_end].copy()),
                                tol=1e-11, maxiter=400)
        u_ref = spl.spsolve(sp.csc_matrix(A), b)
        cg_err = float(np.max(np.abs(u.numpy() - u_ref[be.row_begin:be.row_end])) / np.max(np.abs(u_ref)))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, err, cg_err, res))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, f"{e}\n{traceback.format_exc()}", 0.0, 0.0))


def test_mirrored_form_needs_no_collective_gloo():
    """exchange='none' (the default of build_shard): SpMV without any collective; the
    solver loop on top only all-gathers its vectors"""
    import torch.multiprocessing as mp
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mirror_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, err, cg_err, rr in res:
        assert not isinstance(err, str), f"rank {rank} failed: {err}"
        assert err <= 1e-10 and cg_err <= 1e-9 and rr <= 1e-10, (rank, err, cg_err, rr)
