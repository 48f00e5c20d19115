// cfs_csr.hpp -- the general CSR path (Format::csr): y = A x over EVERY stored entry.
//
// Reference: cpu_mv / cpu_mv_serial, include/matrix/csr_matrix.tpp:2664-2704 -- the kernel
// the reference binds for Format::csr, the ground truth of its own self-check
// (test/test_spmv_mmf.cpp:85-89), and what src/csr.cpp falls back to when a symmetric
// matrix cannot be scheduled.  Two kernel forms (a workgroup per block of rows / a wave per
// chunk of rows), the faster measured at a handle's first SpMV; the block form with an
// XCD-aware block order and, for banded / stencil row blocks, 16-bit column codes and a
// copy of the values in lane order.  Included by cfs_hip.hip (needs HostStage, ensure_init).
#pragma once


// general CSR, streaming form: a workgroup owns a block of consecutive rows whose
// nonzeros fit its LDS product buffer.  All of the block's colind / values loads
// are issued at once (16 per thread, non-temporal: the matrix is read once per
// SpMV), then all x gathers, then the products go to LDS -- three dependent round
// trips per 4 096 nonzeros; the row pointers of the block are staged in LDS
// alongside, so that the row sums (CSR_RT lanes per row, stored order) touch LDS
// only.  Rows longer than the buffer get a block of their own and fall back to
// a whole-workgroup strided sum.
constexpr int kCsrNnz = 4096;  // products per workgroup (32 KiB fp64)
constexpr int kCsrRows = 1024; // rows per block (row pointers in LDS)
// LW: a lane loads LW consecutive entries at once (2: 16-byte value loads of doubles, 8-byte
// column loads -- fewer, wider load instructions for the same bytes).  The gathers of a wave then
// walk every LW-th entry: LW = 4 touches twice as many cache lines per gather instruction as
// LW = 2 and is 5-19 % SLOWER on every stand-in, fp64 and fp32 (profiles/r03_experiment_notes.md)
// -- not instantiated.
template <typename V, int LW = 1>
__global__ void __launch_bounds__(256)
    cfs_csr_stream_kernel(const int32_t *__restrict__ blk_row, int nblocks,
                          const int32_t *__restrict__ rowptr,
                          const int32_t *__restrict__ colind, const V *__restrict__ values,
                          const V *__restrict__ x, V *__restrict__ y, int per_xcd,
                          const uint16_t *__restrict__ col16, const int4 *__restrict__ cbase,
                          const V *__restrict__ vperm, const int32_t *__restrict__ col32) {
  __shared__ V prod[kCsrNnz];
  __shared__ V part[256];
  __shared__ int32_t rps[kCsrRows + 1];
  const int tid = threadIdx.x;
  constexpr int PER = kCsrNnz / 256;
  typedef V VL __attribute__((ext_vector_type(LW > 1 ? LW : 2)));
  typedef int IL __attribute__((ext_vector_type(LW > 1 ? LW : 2)));
  typedef uint16_t HL __attribute__((ext_vector_type(LW > 1 ? LW : 2)));
  // per_xcd > 0: workgroup g runs on XCD g % 8 (the grid is a multiple of 8), and XCD k walks
  // the k-th EIGHTH of the row blocks front to back: the x window of the rows in flight is
  // then fetched into one L2 instead of all eight
  const int nloop = per_xcd > 0 ? per_xcd * 8 : nblocks;
  for (int bb = blockIdx.x; bb < nloop; bb += gridDim.x) {
    const int b = per_xcd > 0 ? (bb & 7) * per_xcd + (bb >> 3) : bb;
    if (b >= nblocks) continue;
    const int r0 = blk_row[b], r1 = blk_row[b + 1];
    const int p0 = rowptr[r0], p1 = rowptr[r1];
    const int n = p1 - p0;
    if (n == 0) { // empty rows only: nothing to load (the clamped loads below would read past the last entry)
      for (int r = r0 + tid; r < r1; r += 256) y[r] = V(0);
      continue;
    }
    if (n <= kCsrNnz) {
      const int nr = r1 - r0;
      // a block whose columns fit four windows of 16 384 has them as 16-bit codes (window << 14 |
      // offset; written once by cfs_csr_narrow_kernel): 2 bytes per nonzero instead of 4
      const int4 cb4 = cbase ? cbase[b] : make_int4(-1, 0, 0, 0);
      const int cb = cb4.x;
      auto decode = [&](uint32_t h) {
        const uint32_t k = h >> 14;
        return (k == 0 ? cb4.x : k == 1 ? cb4.y : k == 2 ? cb4.z : cb4.w) + (int)(h & 0x3fffu);
      };
      int rpv[kCsrRows / 256 + 1];
#pragma unroll
      for (int u = 0; u <= kCsrRows / 256; ++u) rpv[u] = rowptr[r0 + min(tid + u * 256, nr)];
      if (LW > 1) {
        constexpr int PW = PER / LW;
        VL vl[PW];
        IL cl[PW];
        const int qmax = max(n - 1, 0) / LW * LW; // first entry of the last vector (the arrays are padded)
        if (cb >= 0) {
          // A narrow block reads its OWN copy of the values and its 16-bit column codes, both stored
          // in the order the lanes consume them (slot b of 4 096 entries each, written once by
          // cfs_csr_narrow_kernel, zero-padded): the pair a lane loads at position 2 (tid + 256 u)
          // holds entries 512 u + tid and 512 u + 256 + tid -- wide loads, and still every gather
          // instruction of a wave walks 64 CONSECUTIVE entries (the pairs of the natural order
          // walk every second one: twice the cache lines per instruction).  No tail tests: the
          // padding multiplies 0 by x of the block's first window.
          const V *vp = vperm + (size_t)b * kCsrNnz;
          const uint16_t *cp = col16 + (size_t)b * kCsrNnz;
          const int umax = (n - 1) >> 9;
          uint32_t hw[PW];
#pragma unroll
          for (int u = 0; u < PW; ++u) {
            const int q = 2 * (tid + min(u, umax) * 256);
            vl[u] = __builtin_nontemporal_load(reinterpret_cast<const VL *>(vp + q));
            hw[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(cp + q));
          }
#pragma unroll
          for (int u = 0; u <= kCsrRows / 256; ++u)
            if (tid + u * 256 <= nr) rps[tid + u * 256] = rpv[u] - p0;
          V xa[PW], xb[PW];
#pragma unroll
          for (int u = 0; u < PW; ++u) {
            xa[u] = x[decode(hw[u] & 0xffffu)];
            xb[u] = x[decode(hw[u] >> 16)];
          }
#pragma unroll
          for (int u = 0; u < PW; ++u) {
            prod[u * 512 + tid] = vl[u][0] * xa[u];
            prod[u * 512 + 256 + tid] = vl[u][1] * xb[u];
          }
        } else if (cb == -2) {
          // a block that needs more than four windows: the same lane order, 32-bit columns (col32)
          const V *vp = vperm + (size_t)b * kCsrNnz;
          const int32_t *cp = col32 + (size_t)b * kCsrNnz;
          const int umax = (n - 1) >> 9;
#pragma unroll
          for (int u = 0; u < PW; ++u) {
            const int q = 2 * (tid + min(u, umax) * 256);
            vl[u] = __builtin_nontemporal_load(reinterpret_cast<const VL *>(vp + q));
            cl[u] = __builtin_nontemporal_load(reinterpret_cast<const IL *>(cp + q));
          }
#pragma unroll
          for (int u = 0; u <= kCsrRows / 256; ++u)
            if (tid + u * 256 <= nr) rps[tid + u * 256] = rpv[u] - p0;
          V xa[PW], xb[PW];
#pragma unroll
          for (int u = 0; u < PW; ++u) {
            xa[u] = x[cl[u][0]];
            xb[u] = x[cl[u][1]];
          }
#pragma unroll
          for (int u = 0; u < PW; ++u) {
            prod[u * 512 + tid] = vl[u][0] * xa[u];
            prod[u * 512 + 256 + tid] = vl[u][1] * xb[u];
          }
        } else {
#pragma unroll
          for (int u = 0; u < PW; ++u) {
            const int q = min(LW * (tid + u * 256), qmax);
            vl[u] = __builtin_nontemporal_load(reinterpret_cast<const VL *>(values + p0 + q));
            cl[u] = __builtin_nontemporal_load(reinterpret_cast<const IL *>(colind + p0 + q));
          }
#pragma unroll
          for (int u = 0; u <= kCsrRows / 256; ++u)
            if (tid + u * 256 <= nr) rps[tid + u * 256] = rpv[u] - p0;
          V xl[PW][LW];
#pragma unroll
          for (int u = 0; u < PW; ++u) {
            const int q = min(LW * (tid + u * 256), qmax);
#pragma unroll
            for (int j = 0; j < LW; ++j) // (an entry behind the block's last one is not ours: any valid column)
              xl[u][j] = x[q + j < n ? cl[u][j] : cl[u][0]];
          }
#pragma unroll
          for (int u = 0; u < PW; ++u) {
            const int i = LW * (tid + u * 256);
#pragma unroll
            for (int j = 0; j < LW; ++j)
              if (i + j < n) prod[i + j] = vl[u][j] * xl[u][j];
          }
        }
      } else {
        V v[PER];
        int c[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
          const int q = min(tid + u * 256, max(n, 1) - 1);
          v[u] = __builtin_nontemporal_load(values + p0 + q);
          c[u] = __builtin_nontemporal_load(colind + p0 + q);
        }
#pragma unroll
        for (int u = 0; u <= kCsrRows / 256; ++u)
          if (tid + u * 256 <= nr) rps[tid + u * 256] = rpv[u] - p0;
        V xx[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) xx[u] = x[c[u]];
#pragma unroll
        for (int u = 0; u < PER; ++u)
          if (tid + u * 256 < n) prod[tid + u * 256] = v[u] * xx[u];
      }
      __syncthreads();
      // 4 lanes per row, rows strided over the workgroup
      const int sub = tid & 3;
      for (int r = tid >> 2; r < nr; r += 64) {
        const int b0 = rps[r], e0 = rps[r + 1];
        V acc = V(0);
        for (int j = b0 + sub; j < e0; j += 4) acc += prod[j];
        acc += __shfl_down(acc, 2, 4);
        acc += __shfl_down(acc, 1, 4);
        if (sub == 0) y[r0 + r] = acc;
      }
      __syncthreads();
    } else { // one long row (r1 == r0 + 1 by construction)
      V acc = V(0);
      for (int j = p0 + tid; j < p1; j += 256) acc = fma(values[j], x[colind[j]], acc);
      part[tid] = acc;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) part[tid] += part[tid + o];
        __syncthreads();
      }
      if (tid == 0) y[r0] = part[0];
      __syncthreads();
    }
  }
}

// once per handle: the 16-bit column codes of the block form.  The columns of a row block are
// covered greedily by up to four windows of 16 384 columns (window k starts at the smallest
// column not inside windows 0..k-1: banded matrices need one or two, the three planes of a 3-D
// stencil three); a column becomes (window << 14 | offset), the four window starts go to
// cbase[b].  A block that needs more windows (or one long row) keeps reading colind:
// cbase[b].x = -1.
// The codes and a copy of the values of such a block go to slot b (4 096 entries, zero-padded) of
// col16 / vperm IN THE ORDER THE LANES OF cfs_csr_stream_kernel CONSUME THEM: entry 512 U + 256 h + t
// of the block at position 512 U + 2 t + h.
constexpr int kCsrWinBits = 14;
template <typename V>
__global__ void __launch_bounds__(256)
    cfs_csr_narrow_kernel(const int32_t *__restrict__ blk_row, int nblocks, const int32_t *__restrict__ rowptr,
                          const int32_t *__restrict__ colind, const V *__restrict__ values,
                          uint16_t *__restrict__ col16, V *__restrict__ vperm, int4 *__restrict__ cbase,
                          unsigned long long *__restrict__ narrow_nnz, int32_t *__restrict__ col32) {
  __shared__ int smin[256];
  const int tid = threadIdx.x;
  constexpr int PER = kCsrNnz / 256, kNone = 0x7fffffff;
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int p0 = rowptr[blk_row[b]], p1 = rowptr[blk_row[b + 1]];
    const int n = p1 - p0;
    if (n > kCsrNnz || n <= 0) {
      if (tid == 0) cbase[b] = make_int4(-1, 0, 0, 0);
      continue;
    }
    int c[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) c[u] = tid + u * 256 < n ? colind[p0 + tid + u * 256] : kNone;
    int base[4], bound = -1; // columns < bound are covered
    bool more = true;        // (uniform) columns at or beyond `bound` exist
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int lo = kNone;
#pragma unroll
      for (int u = 0; u < PER; ++u)
        if (c[u] >= bound) lo = min(lo, c[u]);
      smin[tid] = lo;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) smin[tid] = min(smin[tid], smin[tid + o]);
        __syncthreads();
      }
      lo = smin[0];
      __syncthreads();
      more = lo != kNone;
      base[k] = more ? lo : (k ? base[k - 1] : 0);
      if (more) bound = lo > kNone - (1 << kCsrWinBits) ? kNone : lo + (1 << kCsrWinBits);
    }
    // anything left beyond the fourth window?
    int left = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) left |= (c[u] != kNone && c[u] >= bound) ? 1 : 0;
    const bool narrow = __syncthreads_or(left) == 0 && base[0] >= 0;
    if (tid == 0) {
      // (x = -2: more than four windows -- lane order too, with 32-bit columns)
      cbase[b] = narrow ? make_int4(base[0], base[1], base[2], base[3]) : make_int4(col32 ? -2 : -1, 0, 0, 0);
      if (narrow) atomicAdd(narrow_nnz, (unsigned long long)n);
      else if (col32) atomicAdd(narrow_nnz + 1, (unsigned long long)n);
    }
    if (narrow) {
      uint16_t *c16 = col16 + (size_t)b * kCsrNnz;
      V *vp = vperm + (size_t)b * kCsrNnz;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int pos = (u >> 1) * 512 + 2 * tid + (u & 1);
        if (tid + u * 256 < n) {
          // the LAST window that starts at or below the column (window starts ascend; unused ones repeat)
          int k = 0;
          if (c[u] >= base[1] && base[1] > base[0]) k = 1;
          if (c[u] >= base[2] && base[2] > base[1]) k = 2;
          if (c[u] >= base[3] && base[3] > base[2]) k = 3;
          c16[pos] = (uint16_t)((k << kCsrWinBits) | (c[u] - base[k]));
          vp[pos] = values[p0 + tid + u * 256];
        } else { // padding: 0 x (x of the first window's first column)
          c16[pos] = 0;
          vp[pos] = V(0);
        }
      }
    } else if (col32) {
      int32_t *c32 = col32 + (size_t)b * kCsrNnz;
      V *vp = vperm + (size_t)b * kCsrNnz;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int pos = (u >> 1) * 512 + 2 * tid + (u & 1);
        const bool in = tid + u * 256 < n;
        c32[pos] = in ? c[u] : base[0]; // padding: 0 x (x of the block's smallest column)
        vp[pos] = in ? values[p0 + tid + u * 256] : V(0);
      }
    }
  }
}

// general CSR, wave-stream form: every WAVE walks its own sequence of chunks
// -- whole rows, at most kCwNnz nonzeros and kCwRows rows, cut on the host: a 16-byte
// descriptor per chunk instead of row-pointer searches -- with no workgroup barrier at
// all.  A wave keeps the NEXT chunk's colind / values loads in flight (a second register
// set, unconditional clamped loads: exact vmcnt bookkeeping) while it gathers x for the
// current one, stages the products in its private 4 KiB of LDS and sums the rows with
// 64 / rows lanes each: the 16-20 resident waves of a CU are each in a different phase
// and their streams never stop.  Rows longer than a chunk are summed by
// cfs_csr_longrow_kernel, a workgroup each.  Which of the two forms a handle launches is
// MEASURED when it is created (five SpMVs each): on the Flan stand-in they are within 5 % of
// each other and the order changes from box to box (block / wave: 279 / 297 us on one,
// 296 / 281 us on another).  (Reference: cpu_mv, csr_matrix.tpp:2683-2704.)
constexpr int kCwNnz = 1024; // products per chunk (16 per lane: 12 KiB of loads in flight per wave)
constexpr int kCwRows = 63;  // rows per chunk (row pointers: one per lane, + the end)
template <typename V>
__global__ void __launch_bounds__(256)
    cfs_csr_wave_kernel(const int4 *__restrict__ cd, int nchunks, const int32_t *__restrict__ rowptr,
                        const int32_t *__restrict__ colind, const V *__restrict__ values,
                        const V *__restrict__ x, V *__restrict__ y) {
  __shared__ V prod[4][kCwNnz];
  constexpr int PER = kCwNnz / 64;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  V *pl = prod[w];
  const int nw = gridDim.x * 4;
  int k = blockIdx.x * 4 + w;
  if (k >= nchunks) return;
  // descriptors {first row, rows, first nonzero, nonzeros} are scalar loads issued TWO chunks
  // ahead: the loads of the next chunk never wait for their descriptor
  int4 dn = cd[k], dnn = cd[min(k + nw, nchunks - 1)];
  V vn[PER];
  int cn[PER], rpn;
  auto fetch = [&](const int4 &dsc) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int q = dsc.z + min(lane + u * 64, max(dsc.w, 1) - 1);
      vn[u] = __builtin_nontemporal_load(values + q);
      cn[u] = __builtin_nontemporal_load(colind + q);
    }
    rpn = rowptr[dsc.x + min(lane, dsc.y)] - dsc.z;
  };
  fetch(dn);
  while (true) {
    const int4 dc = dn;
    V v[PER];
    int c[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) v[u] = vn[u], c[u] = cn[u];
    const int rp = rpn;
    const int kn = k + nw;
    dn = dnn; // (the last chunk once more at the end of a wave's walk: unconditional loads)
    dnn = cd[min(kn + nw, nchunks - 1)];
    fetch(dn);
    V xx[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) xx[u] = x[c[u]];
#pragma unroll
    for (int u = 0; u < PER; ++u)
      if (lane + u * 64 < dc.w) pl[lane + u * 64] = v[u] * xx[u];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // rows: 64 / (rows rounded up to a power of two) lanes each
    int lpr = 64;
    while (lpr > 1 && 64 / lpr < dc.y) lpr >>= 1;
    const int r = lane / lpr, sub = lane & (lpr - 1);
    const int b0 = __shfl(rp, min(r, 63)), e0 = __shfl(rp, min(r + 1, 63));
    V acc = V(0);
    if (r < dc.y)
      for (int j = b0 + sub; j < e0; j += lpr) acc += pl[j];
    for (int o = lpr >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (sub == 0 && r < dc.y) y[dc.x + r] = acc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (kn >= nchunks) break;
    k = kn;
  }
}
// rows longer than a chunk: one workgroup per row
template <typename V>
__global__ void __launch_bounds__(256)
    cfs_csr_longrow_kernel(const int32_t *__restrict__ rows, const int32_t *__restrict__ rowptr,
                           const int32_t *__restrict__ colind, const V *__restrict__ values,
                           const V *__restrict__ x, V *__restrict__ y) {
  __shared__ V part[256];
  const int tid = threadIdx.x, r = rows[blockIdx.x];
  V acc = V(0);
  for (int j = rowptr[r] + tid; j < rowptr[r + 1]; j += 256) acc = fma(values[j], x[colind[j]], acc);
  part[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) part[tid] += part[tid + o];
    __syncthreads();
  }
  if (tid == 0) y[r] = part[0];
}


struct cfs_hip_csr_s {
  int value_bytes = 8, nrows = 0, ncols = 0, nblocks = 0;
  int64_t nnz = 0;
  DevBuf rowptr, colind, values, blk_row;
  DevBuf col16, cbase, vperm; // block form: 16-bit column codes + values of the narrow blocks in lane order, four window starts per block (x = -1: wide)
  int64_t narrow_nnz = 0, wide_lane_nnz = 0; // nonzeros of narrow blocks / of wide blocks kept in lane order
  DevBuf col32;                             // 32-bit columns of the wide blocks, lane order
  DevBuf chunks, longrows; // wave-stream form: chunk descriptors, rows longer than a chunk
  int nchunks = 0, nlong = 0, wave_grid = 0, block_grid = 256 * 8;
  bool xcd_map = true;
  bool form_measured = false; // the faster of the two kernel forms has been chosen (first SpMV)
  int wide = 2;               // block form: entries per lane and load, 1 / 2 (CFS_HIP_CSR_WIDE)
  bool block_form = false; // CFS_HIP_CSR_KERNEL=block: the workgroup-per-block kernel (A/B)
  HostStage stage;
  int device = 0;
};

template <typename V>
static int csr_create(int nrows, int ncols, const int *rowptr, const int *colind,
                      const V *values, cfs_hip_csr_t *out) {
  if (!out || !rowptr || nrows < 0) return set_err(CFS_HIP_ERR_ARG, "bad argument");
  int rc = ensure_init();
  if (rc) return rc;
  int cur_dev = 0;
  HIPCHK(hipGetDevice(&cur_dev));
  auto *m = new cfs_hip_csr_s();
  m->value_bytes = (int)sizeof(V);
  m->device = cur_dev;
  m->nrows = nrows;
  m->ncols = ncols;
  m->nnz = rowptr[nrows];
  if ((rc = m->rowptr.upload(rowptr, ((size_t)nrows + 1) * 4)) ||
      (rc = m->colind.upload(colind, (size_t)m->nnz * 4)) ||
      (rc = m->values.upload(values, (size_t)m->nnz * sizeof(V)))) {
    delete m;
    return rc;
  }
  { // row blocks of at most kCsrNnz nonzeros (a longer row is a block by itself)
    std::vector<int32_t> blk(1, 0);
    int r = 0;
    while (r < nrows) {
      int e = r;
      while (e < nrows && rowptr[e + 1] - rowptr[r] <= kCsrNnz && e - r < kCsrRows) e++;
      if (e == r) e = r + 1;
      blk.push_back(e);
      r = e;
    }
    m->nblocks = (int)blk.size() - 1;
    if ((rc = m->blk_row.upload(blk.data(), blk.size() * 4))) {
      delete m;
      return rc;
    }
    const char *e16 = getenv("CFS_HIP_CSR_COL16");
    if (m->nblocks > 0 && m->nnz > 0 && !(e16 && atoi(e16) == 0)) {
      DevBuf cnt;
      // (the lane-ordered copies are an optimisation: without the memory for them the handle works
      // from the natural-order arrays alone)
      const bool have = !m->col16.alloc((size_t)m->nblocks * kCsrNnz * 2 + 64) && !m->cbase.alloc((size_t)m->nblocks * 16) &&
                        !m->vperm.alloc((size_t)m->nblocks * kCsrNnz * sizeof(V) + 64) &&
                        !m->col32.alloc((size_t)m->nblocks * kCsrNnz * 4 + 64) && !cnt.alloc(16);
      if (!have) (void)hipGetLastError();
      hipStream_t st = cfs_rt::home_stream();
      const char *e32 = getenv("CFS_HIP_CSR_LANE32"); // 0: blocks with 32-bit columns stay in natural order (A/B)
      const bool lane32 = !(e32 && atoi(e32) == 0);
      unsigned long long nn[2] = {0, 0};
      bool ok = have && hipMemsetAsync(cnt.p, 0, 16, st) == hipSuccess;
      if (ok) {
        hipLaunchKernelGGL((cfs_csr_narrow_kernel<V>), dim3(std::min(m->nblocks, 4096)), dim3(256), 0, st,
                           (const int32_t *)m->blk_row.p, m->nblocks, (const int32_t *)m->rowptr.p,
                           (const int32_t *)m->colind.p, (const V *)m->values.p, (uint16_t *)m->col16.p,
                           (V *)m->vperm.p, (int4 *)m->cbase.p, (unsigned long long *)cnt.p,
                           lane32 ? (int32_t *)m->col32.p : (int32_t *)nullptr);
        ok = hipGetLastError() == hipSuccess &&
             hipMemcpyAsync(nn, cnt.p, 16, hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess;
      }
      m->narrow_nnz = (int64_t)nn[0];
      m->wide_lane_nnz = (int64_t)nn[1];
      if (!ok) {
        m->col16 = DevBuf();
        m->cbase = DevBuf();
        m->vperm = DevBuf();
        m->col32 = DevBuf();
        m->narrow_nnz = m->wide_lane_nnz = 0;
      } else { // (an array no block uses is not kept)
        if (m->narrow_nnz == 0) m->col16 = DevBuf();
        if (m->wide_lane_nnz == 0) m->col32 = DevBuf();
      }
      if (getenv("CFS_PLAN_VERBOSE"))
        fprintf(stderr, "[cfs_hip] general CSR: %lld of %lld nonzeros in blocks with 16-bit columns, %lld in lane-ordered blocks with 32-bit columns\n",
                (long long)m->narrow_nnz, (long long)m->nnz, (long long)m->wide_lane_nnz);
    }
  }
  { // wave-stream form: chunks of whole rows (<= kCwNnz nonzeros, <= kCwRows rows); longer rows apart
    std::vector<int4> cd;
    std::vector<int32_t> lr;
    int r = 0;
    while (r < nrows) {
      if (rowptr[r + 1] - rowptr[r] > kCwNnz) {
        lr.push_back(r++);
        continue;
      }
      int e = r;
      while (e < nrows && e - r < kCwRows && rowptr[e + 1] - rowptr[r] <= kCwNnz) e++;
      cd.push_back(make_int4(r, e - r, rowptr[r], rowptr[e] - rowptr[r]));
      r = e;
    }
    m->xcd_map = true; // XCD k takes the k-th eighth of the matrix (Flan stand-in: x is then read ~once, not 8 times)
    if (const char *e = getenv("CFS_HIP_CSR_XCD")) m->xcd_map = atoi(e) != 0;
    if (m->xcd_map && !cd.empty()) {
      // the same for the wave form, by the ORDER of the descriptors: workgroup g (4 waves, XCD
      // g % 8) reads descriptors 4g .. 4g+3; empty descriptors fill the last eighth
      const int ng = ((int)cd.size() + 3) / 4, per = (ng + 7) / 8;
      std::vector<int4> perm((size_t)per * 32, make_int4(0, 0, 0, 0));
      for (int g = 0; g < per * 8; g++)
        for (int w = 0; w < 4; w++) {
          const size_t phys = ((size_t)(g & 7) * per + (g >> 3)) * 4 + w;
          if (phys < cd.size()) perm[(size_t)g * 4 + w] = cd[phys];
        }
      cd.swap(perm);
    }
    m->nchunks = (int)cd.size();
    m->nlong = (int)lr.size();
    if ((rc = m->chunks.upload(cd.data(), cd.size() * sizeof(int4))) ||
        (rc = m->longrows.upload(lr.data(), lr.size() * 4))) {
      delete m;
      return rc;
    }
    // persistent waves: as many workgroups as are co-resident
    int nb = 0;
    const void *k = sizeof(V) == 8 ? (const void *)cfs_csr_wave_kernel<double> : (const void *)cfs_csr_wave_kernel<float>;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, 0) != hipSuccess || nb < 1) nb = 4;
    if (hipGetDeviceProperties(&prop, m->device) != hipSuccess) prop.multiProcessorCount = 256;
    m->wave_grid = std::max(1, std::min((m->nchunks + 3) / 4, prop.multiProcessorCount * nb));
    if (m->xcd_map) m->wave_grid = std::max(8, m->wave_grid & ~7);
    m->block_form = true;
    m->form_measured = false;
    m->wide = 2; // pairs of entries per lane (Flan stand-in fp64, against single entries: 275-289 us against 284-294)
    if (const char *e = getenv("CFS_HIP_CSR_WIDE")) m->wide = atoi(e) >= 2 ? 2 : 1;
    if (const char *e = getenv("CFS_HIP_CSR_KERNEL")) { // block | wave: no measurement
      m->block_form = strcmp(e, "wave") != 0;
      m->form_measured = true;
    }
  }
  *out = m;
  return 0;
}

static int csr_launch(cfs_hip_csr_t h, void *y, const void *x, hipStream_t st);
// the faster of the two kernel forms, measured once with the caller's own vectors (y is fully
// overwritten by either): five SpMVs each after one warm-up, on the caller's stream
static int csr_choose_form(cfs_hip_csr_t h, void *y, const void *x, hipStream_t st) {
  h->form_measured = true;
  if (h->nnz < (int64_t)1 << 20) return 0; // small: the block form
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return 0;
  float t[2] = {0, 0};
  for (int form = 0; form < 2; form++) {
    h->block_form = form == 0;
    int rc = csr_launch(h, y, x, st);
    (void)hipEventRecord(a, st);
    for (int it = 0; it < 5 && !rc; it++) rc = csr_launch(h, y, x, st);
    (void)hipEventRecord(b, st);
    if (rc || hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&t[form], a, b) != hipSuccess) t[form] = 1e30f;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  h->block_form = t[0] <= t[1];
  if (getenv("CFS_PLAN_VERBOSE"))
    fprintf(stderr, "[cfs_hip] general CSR kernel: block form %.1f us, wave form %.1f us\n", t[0] * 200.0, t[1] * 200.0);
  return 0;
}

static int csr_launch(cfs_hip_csr_t h, void *y, const void *x, hipStream_t st) {
  if (!h->block_form) {
    if (h->nchunks > 0) {
      if (h->value_bytes == 8)
        hipLaunchKernelGGL((cfs_csr_wave_kernel<double>), dim3(h->wave_grid), dim3(256), 0, st,
                           (const int4 *)h->chunks.p, h->nchunks, (const int32_t *)h->rowptr.p,
                           (const int32_t *)h->colind.p, (const double *)h->values.p, (const double *)x, (double *)y);
      else
        hipLaunchKernelGGL((cfs_csr_wave_kernel<float>), dim3(h->wave_grid), dim3(256), 0, st,
                           (const int4 *)h->chunks.p, h->nchunks, (const int32_t *)h->rowptr.p,
                           (const int32_t *)h->colind.p, (const float *)h->values.p, (const float *)x, (float *)y);
    }
    if (h->nlong > 0) {
      if (h->value_bytes == 8)
        hipLaunchKernelGGL((cfs_csr_longrow_kernel<double>), dim3(h->nlong), dim3(256), 0, st,
                           (const int32_t *)h->longrows.p, (const int32_t *)h->rowptr.p, (const int32_t *)h->colind.p,
                           (const double *)h->values.p, (const double *)x, (double *)y);
      else
        hipLaunchKernelGGL((cfs_csr_longrow_kernel<float>), dim3(h->nlong), dim3(256), 0, st,
                           (const int32_t *)h->longrows.p, (const int32_t *)h->rowptr.p, (const int32_t *)h->colind.p,
                           (const float *)h->values.p, (const float *)x, (float *)y);
    }
  } else if (h->nblocks > 0) {
    const int per_xcd = h->xcd_map ? (h->nblocks + 7) / 8 : 0;
    const int want = h->xcd_map ? per_xcd * 8 : h->nblocks;
    const int grid = want < 256 * 8 ? want : 256 * 8;
#define CFS_CSR_BLOCK(V, W)                                                                              \
  hipLaunchKernelGGL((cfs_csr_stream_kernel<V, W>), dim3(grid), dim3(256), 0, st, (const int32_t *)h->blk_row.p, \
                     h->nblocks, (const int32_t *)h->rowptr.p, (const int32_t *)h->colind.p,                \
                     (const V *)h->values.p, (const V *)x, (V *)y, per_xcd, (const uint16_t *)h->col16.p,  \
                     (const int4 *)(h->wide == 2 ? h->cbase.p : nullptr), (const V *)h->vperm.p,                \
                     (const int32_t *)h->col32.p)
    if (h->value_bytes == 8) {
      if (h->wide == 2) CFS_CSR_BLOCK(double, 2);
      else CFS_CSR_BLOCK(double, 1);
    } else {
      if (h->wide == 2) CFS_CSR_BLOCK(float, 2);
      else CFS_CSR_BLOCK(float, 1);
    }
#undef CFS_CSR_BLOCK
  }
  HIPCHK(hipGetLastError());
  return 0;
}

