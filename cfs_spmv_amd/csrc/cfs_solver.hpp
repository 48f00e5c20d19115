// cfs_solver.hpp -- a solver-style caller of the SpMV path behind the C ABI: conjugate
// gradients on resident vectors.
//
// The reference has no counterpart: its only callers are a benchmark loop and a
// self-check with a fixed x (bench/bench_spmv_mmf.cpp:139-173, test/test_spmv_mmf.cpp:71-109).
// A solver feeds every product back as the next input; on a GPU the loop must not come
// back to the host between products -- a dot product read on the host costs a stream
// synchronisation (20-30 us), more than a whole SpMV of a small matrix.  Here an iteration
// is five launches on one stream and no host round trip:
//   tile kernel + fold            q = A p                        (cfs_hip.hip)
//   cg_pq_kernel                  pq = p . q
//   cg_update_kernel              u += (rr/pq) p;  r -= (rr/pq) q;  rr' = r . r   (one pass)
//   cg_direction_kernel           p = r + (rr'/rr) p;  converged?
// The scalars live in device memory as 512 partial sums each (one per workgroup of the kernel
// that produced them, fp64 whatever the value type); every workgroup of the consuming kernel adds
// them up for itself in a fixed order -- no atomics, bit-reproducible scalars (with a
// deterministic handle the whole solve is).  r . r has two slots, alternating with the parity of
// the iteration (the host passes the iteration number as a kernel argument).  The host looks at
// the convergence flag every `check_every` iterations (one 8-byte copy + synchronisation);
// kernels enqueued behind a converged iteration return at once.  cfs_hip_sym_cg recomputes the
// true residual ||b - A u|| / ||b|| at the end.
#pragma once

namespace cfs_solver {

// partial sums: one double per workgroup and quantity (no atomics: thousands of device-scope atomics
// on one address cost tens of microseconds -- they are resolved one after the other -- and their
// order would vary; a fixed grid and a fixed summation order make the scalars bit-reproducible)
constexpr int kThreads = 256;
constexpr int kGrid = 512;                                  // workgroups of every vector kernel
enum Slot { P_RR0 = 0, P_RR1, P_PQ, P_BB, P_RES, P_COUNT }; // part[slot][kGrid]
enum Counter { I_ITER = 0, I_DONE, I_COUNT };

// sum of v over the workgroup, returned to every thread (fixed order)
__device__ __forceinline__ double block_sum(double v) {
  __shared__ double part[kThreads / 64];
  __shared__ double total;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads(); // (a second call may not overwrite `total` before everybody has read it)
  if (lane == 0) part[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) s += part[w];
    total = s;
  }
  __syncthreads();
  return total;
}
// the scalar a slot holds: the sum of its kGrid partial sums, by every workgroup for itself
__device__ __forceinline__ double slot_sum(const double *part, int slot) {
  double v = 0.0;
  for (int g = threadIdx.x; g < kGrid; g += kThreads) v += part[slot * kGrid + g];
  return block_sum(v);
}

// The vector kernels move 16 bytes per lane and load (two doubles / four floats) over the part of
// the vectors that is a multiple of that -- every vector here is 16-byte aligned: library memory, or
// checked by cg() -- and single values over the rest; grid-stride.
template <typename V> struct Vec16 {
  static constexpr int W = 16 / (int)sizeof(V);
  typedef V type __attribute__((ext_vector_type(16 / sizeof(V))));
};

// r = b - q;  p = r (when given);  part[slot_rr] <- r . r;  part[P_BB] <- b . b (when with_bb)
template <typename V>
__global__ void __launch_bounds__(kThreads)
    cg_residual_kernel(V *__restrict__ r, V *__restrict__ p, const V *__restrict__ b, const V *__restrict__ q,
                       long long n, double *__restrict__ part, int slot_rr, int with_bb) {
  constexpr int W = Vec16<V>::W;
  typedef typename Vec16<V>::type VT;
  const long long nv = n / W, t0 = (long long)blockIdx.x * kThreads + threadIdx.x, stride = (long long)gridDim.x * kThreads;
  double rr = 0.0, bb = 0.0;
  for (long long i = t0; i < nv; i += stride) {
    const VT bv = reinterpret_cast<const VT *>(b)[i], qv = reinterpret_cast<const VT *>(q)[i];
    VT rv;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      rv[k] = bv[k] - qv[k];
      rr += (double)rv[k] * (double)rv[k];
      bb += (double)bv[k] * (double)bv[k];
    }
    if (r) reinterpret_cast<VT *>(r)[i] = rv;
    if (p) reinterpret_cast<VT *>(p)[i] = rv;
  }
  for (long long i = nv * W + t0; i < n; i += stride) {
    const V bi = b[i], ri = bi - q[i];
    if (r) r[i] = ri;
    if (p) p[i] = ri;
    rr += (double)ri * (double)ri;
    bb += (double)bi * (double)bi;
  }
  rr = block_sum(rr);
  bb = block_sum(bb);
  if (threadIdx.x == 0) {
    part[slot_rr * kGrid + blockIdx.x] = rr;
    if (with_bb) part[P_BB * kGrid + blockIdx.x] = bb;
  }
}

// part[P_PQ] <- p . q
template <typename V>
__global__ void __launch_bounds__(kThreads)
    cg_pq_kernel(const V *__restrict__ p, const V *__restrict__ q, long long n, double *__restrict__ part,
                 const int *__restrict__ ic) {
  if (ic[I_DONE]) return;
  constexpr int W = Vec16<V>::W;
  typedef typename Vec16<V>::type VT;
  const long long nv = n / W, t0 = (long long)blockIdx.x * kThreads + threadIdx.x, stride = (long long)gridDim.x * kThreads;
  double s = 0.0;
  for (long long i = t0; i < nv; i += stride) {
    const VT pv = reinterpret_cast<const VT *>(p)[i], qv = reinterpret_cast<const VT *>(q)[i];
#pragma unroll
    for (int k = 0; k < W; ++k) s += (double)pv[k] * (double)qv[k];
  }
  for (long long i = nv * W + t0; i < n; i += stride) s += (double)p[i] * (double)q[i];
  s = block_sum(s);
  if (threadIdx.x == 0) part[P_PQ * kGrid + blockIdx.x] = s;
}

// alpha = rr / pq;  u += alpha p;  r -= alpha q;  part[rr of the next iteration] <- r . r
template <typename V>
__global__ void __launch_bounds__(kThreads)
    cg_update_kernel(V *__restrict__ u, V *__restrict__ r, const V *__restrict__ p, const V *__restrict__ q,
                     long long n, double *__restrict__ part, const int *__restrict__ ic, int it) {
  if (ic[I_DONE]) return;
  const double pq = slot_sum(part, P_PQ), rr_old = slot_sum(part, P_RR0 + (it & 1));
  const double alpha = pq != 0.0 ? rr_old / pq : 0.0;
  constexpr int W = Vec16<V>::W;
  typedef typename Vec16<V>::type VT;
  const long long nv = n / W, t0 = (long long)blockIdx.x * kThreads + threadIdx.x, stride = (long long)gridDim.x * kThreads;
  double s = 0.0;
  for (long long i = t0; i < nv; i += stride) {
    VT uv = reinterpret_cast<VT *>(u)[i], rv = reinterpret_cast<VT *>(r)[i];
    const VT pv = reinterpret_cast<const VT *>(p)[i], qv = reinterpret_cast<const VT *>(q)[i];
#pragma unroll
    for (int k = 0; k < W; ++k) {
      uv[k] = (V)((double)uv[k] + alpha * (double)pv[k]);
      const double ri = (double)rv[k] - alpha * (double)qv[k];
      rv[k] = (V)ri;
      s += ri * ri;
    }
    reinterpret_cast<VT *>(u)[i] = uv;
    reinterpret_cast<VT *>(r)[i] = rv;
  }
  for (long long i = nv * W + t0; i < n; i += stride) {
    u[i] = (V)((double)u[i] + alpha * (double)p[i]);
    const double ri = (double)r[i] - alpha * (double)q[i];
    r[i] = (V)ri;
    s += ri * ri;
  }
  s = block_sum(s);
  if (threadIdx.x == 0) part[(P_RR0 + ((it + 1) & 1)) * kGrid + blockIdx.x] = s;
}

// beta = rr' / rr;  p = r + beta p;  one thread: iteration count, convergence flag (rr' <= stop)
template <typename V>
__global__ void __launch_bounds__(kThreads)
    cg_direction_kernel(V *__restrict__ p, const V *__restrict__ r, long long n, const double *__restrict__ part,
                        int *__restrict__ ic, int it, double stop) {
  if (ic[I_DONE]) return;
  const double rr = slot_sum(part, P_RR0 + (it & 1)), rrn = slot_sum(part, P_RR0 + ((it + 1) & 1));
  const double beta = rr != 0.0 ? rrn / rr : 0.0;
  constexpr int W = Vec16<V>::W;
  typedef typename Vec16<V>::type VT;
  const long long nv = n / W, t0 = (long long)blockIdx.x * kThreads + threadIdx.x, stride = (long long)gridDim.x * kThreads;
  for (long long i = t0; i < nv; i += stride) {
    VT pv = reinterpret_cast<VT *>(p)[i];
    const VT rv = reinterpret_cast<const VT *>(r)[i];
#pragma unroll
    for (int k = 0; k < W; ++k) pv[k] = (V)((double)rv[k] + beta * (double)pv[k]);
    reinterpret_cast<VT *>(p)[i] = pv;
  }
  for (long long i = nv * W + t0; i < n; i += stride) p[i] = (V)((double)r[i] + beta * (double)p[i]);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ic[I_ITER] += 1; // (counted on the device: a replayed graph passes the same `it` again, only its parity matters)
    // (!(x > y): a NaN residual also ends the iteration)
    if (!(rrn > stop)) ic[I_DONE] = 1;
  }
}

// u: in = first guess, out = solution.  Returns 0 / an error code; *iterations, *relres as documented in cfs_hip.h
template <typename V, class Handle>
int cg(Handle *h, void *u_dev, const void *b_dev, double tol, int maxiter, int check_every, int *iterations,
       double *relres, hipStream_t st) {
  using cfs_rt::DevBuf;
  const long long n = h->n();
  if (h->rows() != h->n()) return cfs_rt::set_err(CFS_HIP_ERR_UNSUPPORTED, "cg: the handle holds a row block, not the whole matrix");
  if (maxiter < 0 || !(tol >= 0.0)) return cfs_rt::set_err(CFS_HIP_ERR_ARG, "cg: bad tolerance / iteration limit");
  if (check_every < 1) check_every = 8;
  // (more than ~100 launches enqueued ahead of the GPU make the runtime stall: pwtk stand-in, 64
  // iterations = 320 launches between two looks, 187 us per iteration instead of 32)
  check_every = std::min(check_every, 16);
  const char *eg = getenv("CFS_HIP_CG_GRAPH");
  const bool use_graph = eg && atoi(eg) != 0;
  if ((((uintptr_t)u_dev) | ((uintptr_t)b_dev)) & 15)
    return cfs_rt::set_err(CFS_HIP_ERR_ARG, "cg: u and b must be 16-byte aligned");
  V *u = (V *)u_dev;
  const V *b = (const V *)b_dev;
  DevBuf rbuf, pbuf, qbuf, pbuf_part, cnt;
  int rc;
  if ((rc = rbuf.alloc((size_t)n * sizeof(V) + 64)) || (rc = pbuf.alloc((size_t)n * sizeof(V) + 64)) ||
      (rc = qbuf.alloc((size_t)n * sizeof(V) + 64)) || (rc = pbuf_part.alloc((size_t)P_COUNT * kGrid * sizeof(double))) ||
      (rc = cnt.alloc(I_COUNT * sizeof(int))))
    return rc;
  V *r = (V *)rbuf.p, *p = (V *)pbuf.p, *q = (V *)qbuf.p;
  double *part = (double *)pbuf_part.p;
  int *ic = (int *)cnt.p;
  std::vector<double> hp((size_t)P_COUNT * kGrid);
  auto read_slot = [&](int slot, double *out) -> int { // (synchronises the stream)
    HIPCHK(hipMemcpyAsync(hp.data(), part, hp.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    double s = 0.0;
    for (int g = 0; g < kGrid; g++) s += hp[(size_t)slot * kGrid + g];
    *out = s;
    return 0;
  };
  HIPCHK(hipMemsetAsync(part, 0, (size_t)P_COUNT * kGrid * sizeof(double), st));
  HIPCHK(hipMemsetAsync(ic, 0, I_COUNT * sizeof(int), st));
  // r = b - A u, p = r, rr[0] = r . r, bb = b . b
  if ((rc = h->spmv_local(q, u, nullptr, st))) return rc;
  hipLaunchKernelGGL((cg_residual_kernel<V>), dim3(kGrid), dim3(kThreads), 0, st, r, p, b, (const V *)q, n, part,
                     (int)P_RR0, 1);
  HIPCHK(hipGetLastError());
  double rr0 = 0.0, bb = 0.0;
  if ((rc = read_slot(P_RR0, &rr0))) return rc;
  for (int g = 0; g < kGrid; g++) bb += hp[(size_t)P_BB * kGrid + g];
  const double stop = tol * tol * bb;
  bool done = !(rr0 > stop); // the first guess already solves it (or b = 0)
  int host_ic[I_COUNT] = {0, 0};
  int it = 0;
  auto iteration = [&](int k) -> int {
    int r2 = h->spmv_local(q, p, nullptr, st);
    if (r2) return r2;
    hipLaunchKernelGGL((cg_pq_kernel<V>), dim3(kGrid), dim3(kThreads), 0, st, (const V *)p, (const V *)q, n, part,
                       (const int *)ic);
    hipLaunchKernelGGL((cg_update_kernel<V>), dim3(kGrid), dim3(kThreads), 0, st, u, r, (const V *)p, (const V *)q, n,
                       part, (const int *)ic, k);
    hipLaunchKernelGGL((cg_direction_kernel<V>), dim3(kGrid), dim3(kThreads), 0, st, p, (const V *)r, n,
                       (const double *)part, ic, k, stop);
    return 0;
  };
  // CFS_HIP_CG_GRAPH=1: two iterations (both parities) captured once and replayed as one graph launch.
  // Measured and NOT the default: the replay is 3-10 % slower than ten plain launches (pwtk stand-in
  // 34.2 against 31.1 us per iteration, ldoor 91 / 87, Flan 145 / 142) -- as for the SpMV's two launches
  // alone; the host enqueues plain launches faster than the GPU retires them anyway.
  hipGraph_t graph = nullptr;
  hipGraphExec_t gexec = nullptr;
  struct GraphGuard {
    hipGraph_t &g;
    hipGraphExec_t &e;
    ~GraphGuard() {
      if (e) (void)hipGraphExecDestroy(e);
      if (g) (void)hipGraphDestroy(g);
    }
  } graph_guard{graph, gexec};
  if (use_graph && !done && maxiter >= 2 && st != nullptr) {
    bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
      const int r0 = iteration(0), r1 = r0 ? r0 : iteration(1);
      ok = hipStreamEndCapture(st, &graph) == hipSuccess && !r1 && graph &&
           hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0) == hipSuccess;
    }
    if (!ok) {
      (void)hipGetLastError();
      gexec = nullptr;
    }
  }
  while (!done && it < maxiter) {
    const int until = std::min(maxiter, it + check_every);
    while (it < until) {
      if (gexec && (it & 1) == 0 && it + 2 <= until) {
        HIPCHK(hipGraphLaunch(gexec, st));
        it += 2;
      } else {
        if ((rc = iteration(it))) return rc;
        ++it;
      }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host_ic, ic, sizeof host_ic, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    done = host_ic[I_DONE] != 0;
  }
  // the true residual of what is returned
  if ((rc = h->spmv_local(q, u, nullptr, st))) return rc;
  hipLaunchKernelGGL((cg_residual_kernel<V>), dim3(kGrid), dim3(kThreads), 0, st, (V *)nullptr, (V *)nullptr, b,
                     (const V *)q, n, part, (int)P_RES, 0);
  HIPCHK(hipGetLastError());
  double res2 = 0.0;
  if ((rc = read_slot(P_RES, &res2))) return rc;
  HIPCHK(hipMemcpy(host_ic, ic, sizeof host_ic, hipMemcpyDeviceToHost));
  if (iterations) *iterations = host_ic[I_ITER];
  if (relres) *relres = bb > 0.0 ? std::sqrt(res2 / bb) : std::sqrt(res2);
  return 0;
}

} // namespace cfs_solver
