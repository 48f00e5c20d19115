// cfs_comm.hpp -- the exchange of the off-block y contributions, native (no Python, no
// torch.distributed): the north-star's reduce-scatter over xGMI behind the C ABI.
//
// The reference has no counterpart (it is a single-process OpenMP code whose only
// exchange is the barrier between colours, csr_matrix.tpp:3018); what is exchanged
// here are its DIRECT CONFLICTS (csr_matrix.tpp:1443-1451): transposed updates
// y_j += a_ij x_i whose row j belongs to another row block.
//
// One process drives N devices (CFS_NUM_GPUS behind the C++ surface):
//   * transport "rccl": one communicator per device (ncclCommInitAll), collectives
//     issued for all ranks between ncclGroupStart / ncclGroupEnd, each on its
//     device's stream.  librccl.so is dlopen'ed at the first use: the library has
//     no link-time dependency on it, and a box without RCCL keeps everything else.
//     RCCL refuses two ranks on one device, so
//   * transport "peer": the same reduce-scatter / all-gather as plain kernels and
//     copies over peer access -- what N shards on ONE device (the test boxes) use,
//     and the fall-back when RCCL cannot be loaded.
// Either way: sum-reduce-scatter of N x count values per rank (rank r receives the
// r-th block), and all-gather of count values per rank.
#pragma once

#include <dlfcn.h>

namespace cfs_comm {

using cfs_rt::DeviceGuard;
using cfs_rt::set_err;

// the few RCCL entry points, resolved at run time (signatures: rccl/rccl.h)
struct Rccl {
  void *lib = nullptr;
  int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
  int (*CommDestroy)(void *comm) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*ReduceScatter)(const void *send, void *recv, size_t recvcount, int dtype, int op, void *comm,
                       hipStream_t st) = nullptr;
  int (*AllGather)(const void *send, void *recv, size_t sendcount, int dtype, void *comm, hipStream_t st) = nullptr;
  bool ok = false;
  std::string why;
};
inline Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names)
      if ((r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
    if (!r.lib) {
      r.why = std::string("librccl.so not loadable: ") + (dlerror() ? dlerror() : "?");
      return;
    }
    auto sym = [&](const char *n) { return dlsym(r.lib, n); };
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.ReduceScatter = (decltype(r.ReduceScatter))sym("ncclReduceScatter");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.ReduceScatter && r.AllGather;
    if (!r.ok) r.why = "librccl.so lacks an entry point";
  });
  return r;
}
constexpr int kNcclSum = 0, kNcclFloat32 = 7, kNcclFloat64 = 8; // rccl.h enumerators

// out[i] = sum over ranks g of in[g][off + i]  (the peer transport's reduce-scatter)
template <typename V>
__global__ void __launch_bounds__(256)
    cfs_peer_sum_kernel(V *__restrict__ out, const V *const *__restrict__ in, int nranks, size_t off, size_t count) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    V s = V(0);
    for (int g = 0; g < nranks; ++g) s += in[g][off + i]; // fixed order: bit-reproducible
    out[i] = s;
  }
}

} // namespace cfs_comm

struct cfs_hip_comm_s {
  std::vector<int> dev;
  std::vector<void *> comm;      // RCCL communicators (transport rccl)
  bool use_rccl = false;
  std::string note;              // why the peer transport is in use
  std::vector<cfs_rt::DevBuf> ptrs; // per rank: the table of the ranks' send buffers (peer transport)
  std::vector<hipEvent_t> ready, done; // peer transport: send buffer written / block summed
  bool done_valid = false;
  ~cfs_hip_comm_s() {
    for (size_t g = 0; g < dev.size(); g++) {
      cfs_rt::DeviceGuard dg(dev[g]);
      if (use_rccl && g < comm.size() && comm[g]) (void)cfs_comm::rccl().CommDestroy(comm[g]);
      if (g < ready.size() && ready[g]) (void)hipEventDestroy(ready[g]);
      if (g < done.size() && done[g]) (void)hipEventDestroy(done[g]);
    }
  }
};
