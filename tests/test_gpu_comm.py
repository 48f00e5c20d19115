"""The native exchange of the C ABI (cfs_hip_comm_*, cfs_comm.hpp) and the exchange form of a
one-process multi-device handle (MultiSym::spmv_exchange): the north-star's reduce-scatter of
the off-block y contributions without Python or torch.distributed.

On a one-GPU box: the PEER transport with 2-4 ranks sharing cuda:0 (RCCL refuses two ranks
per device), and the RCCL transport with ONE rank (the N = 1 rehearsal: ncclCommInitAll,
ncclReduceScatter and ncclAllGather really run).  What crosses a block boundary are the
reference's direct conflicts, include/matrix/csr_matrix.tpp:1443-1451."""
import ctypes as C

import numpy as np
import pytest

import cfs_spmv_amd as cfs
from cfs_spmv_amd import _lib, synth
from conftest import scaled_err

pytestmark = pytest.mark.gpu
TOL = {np.float64: 1e-12, np.float32: 1e-5}
AUTO, RCCL, PEER = 0, 1, 2


def _torch():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.init()
    return torch


def _ptrs(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


@pytest.mark.parametrize("transport,nranks", [(PEER, 2), (PEER, 4), (RCCL, 1), (AUTO, 3)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_reduce_scatter_and_allgather(transport, nranks, dtype):
    torch = _torch()
    lib = _lib.load()
    comm = C.c_void_p()
    devs = (C.c_int * nranks)(*([0] * nranks))
    _lib.check(lib.cfs_hip_comm_create(nranks, devs, transport, C.byref(comm)))
    nd, tr = C.c_int(), C.c_int()
    _lib.check(lib.cfs_hip_comm_info(comm, C.byref(nd), C.byref(tr)))
    assert nd.value == nranks
    assert tr.value == (RCCL if (transport == RCCL or (transport == AUTO and nranks == 1)) else PEER)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    count = 1000
    rng = np.random.default_rng(7)
    send_h = [rng.uniform(-1, 1, nranks * count).astype(dtype) for _ in range(nranks)]
    send = [torch.from_numpy(s).cuda() for s in send_h]
    recv = [torch.full((count,), float("nan"), dtype=tdt, device="cuda") for _ in range(nranks)]
    streams = [torch.cuda.Stream() for _ in range(nranks)]
    sp = (C.c_void_p * nranks)(*[s.cuda_stream for s in streams])
    torch.cuda.synchronize()
    for rep in range(3):  # repeated: the send buffers are reused (wait_consumed)
        for g in range(nranks):
            _lib.check(lib.cfs_hip_comm_wait_consumed(comm, g, sp[g]))
        _lib.check(lib.cfs_hip_comm_reduce_scatter(comm, _ptrs(send), _ptrs(recv), count, np.dtype(dtype).itemsize, sp))
    torch.cuda.synchronize()
    total = np.sum(np.stack([s.astype(np.float64) for s in send_h]), axis=0)
    for r in range(nranks):
        ref = total[r * count:(r + 1) * count]
        got = recv[r].cpu().numpy().astype(np.float64)
        assert np.max(np.abs(got - ref)) <= (1e-14 if dtype == np.float64 else 1e-5) * nranks
    # all-gather: every rank ends with every rank's block
    blocks = [torch.from_numpy(rng.uniform(-1, 1, count).astype(dtype)).cuda() for _ in range(nranks)]
    full = [torch.full((nranks * count,), float("nan"), dtype=tdt, device="cuda") for _ in range(nranks)]
    torch.cuda.synchronize()
    _lib.check(lib.cfs_hip_comm_allgather(comm, _ptrs(blocks), _ptrs(full), count, np.dtype(dtype).itemsize, sp))
    torch.cuda.synchronize()
    ref = torch.cat(blocks)
    for r in range(nranks):
        assert torch.equal(full[r], ref)
    _lib.check(lib.cfs_hip_comm_destroy(comm))


def test_rccl_transport_refuses_shared_devices():
    _torch()
    comm = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    rc = _lib.load().cfs_hip_comm_create(2, devs, RCCL, C.byref(comm))
    assert rc == -3  # CFS_HIP_ERR_UNSUPPORTED: RCCL needs one rank per device
    assert b"one rank per device" in _lib.load().cfs_hip_last_error()


@pytest.mark.parametrize("ngpus", [2, 3, 8])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_multi_device_handle_exchange_form(ngpus, dtype, monkeypatch):
    """CFS_HIP_FLAG_SHARD_EXCHANGE at cfs_hip_sym_create_multi_*: exchange-form shards, one
    native reduce-scatter per SpMV (here over the PEER transport: the shards share cuda:0),
    local fold beside it; the handle still behaves like the whole matrix"""
    from oracle import oracle
    torch = _torch()
    n, rp, ci, va, _ = synth.generate("Flan_1565", 0.03)
    va = va.astype(dtype)
    x = synth.make_x(n, 42, dtype)
    A = cfs.SymMatrix(n, rp, ci, va, ngpus=ngpus, options=cfs.make_options(flags=cfs.FLAG_SHARD_EXCHANGE | 32))
    y_ld, absrow = oracle.csr_spmv_ld(n, rp, ci, va, x)
    xd = torch.from_numpy(x).cuda()
    for garbage in (7.0, -1.0, float("nan")):
        yd = torch.full((n,), garbage, dtype=xd.dtype, device="cuda")
        A.dense_vector_multiply(yd, xd)
        torch.cuda.synchronize()
        assert scaled_err(yd.cpu().numpy(), y_ld, absrow) <= TOL[dtype], (ngpus, garbage)
    # with replicated x and local y blocks on top (the copy path of a multi-GPU node)
    _lib.check(_lib.load().cfs_hip_sym_multi_set_xmode(A._h, 2))
    yd = torch.full((n,), 3.0, dtype=xd.dtype, device="cuda")
    A.dense_vector_multiply(yd, xd)
    torch.cuda.synchronize()
    assert scaled_err(yd.cpu().numpy(), y_ld, absrow) <= TOL[dtype]
    A.close()
