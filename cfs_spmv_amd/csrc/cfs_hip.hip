// cfs_hip.hip -- gfx950 (MI355X, CDNA4) kernels and the C ABI of libcfs_hip.so.
//
// The hot path of athelaf/cfs-spmv -- cpu_mv_sym_conflict_free_v2,
// include/matrix/csr_matrix.tpp:2965-3028: walk the strict lower triangle once,
// update both y_i (row side) and y_j (transposed side) -- as hand-written HIP
// for 64-lane wavefronts.  HBM-bandwidth bound (0.3 flop/byte): no MFMA.
// The schedule the kernels walk is built by cfs_plan.hpp.
//
// Written for gfx950 only; compile with hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cfs_hip.h"
#include "cfs_plan.hpp"

using cfs_plan::SymPlan;
using cfs_plan::Tile;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local std::string g_err;
static int set_err(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                          \
  do {                                                                        \
    hipError_t e__ = (expr);                                                  \
    if (e__ != hipSuccess)                                                    \
      return set_err(CFS_HIP_ERR_DEVICE, std::string(#expr) + ": " +          \
                                             hipGetErrorString(e__));         \
  } while (0)

static int g_device = -1;
static hipStream_t g_stream = nullptr;

static int ensure_init() {
  if (g_device >= 0) return 0;
  return cfs_hip_init(0);
}

// ---------------------------------------------------------------------------
// device view of a plan
// ---------------------------------------------------------------------------
template <typename V> struct SymDev {
  const Tile *tiles;
  const int32_t *group_ptr;
  const int32_t *halo_col;
  const uint32_t *rowinfo;
  const V *diag;
  const uint32_t *slice_off;
  const V *vals;
  const uint16_t *slots;
  V *strip;
  int row_begin;
  int lds_slots;
};

template <typename V> struct Vec4;
template <> struct Vec4<double> { double2 a, b; };
template <> struct Vec4<float> { float4 a; };

// one packet = 4 jagged diagonals x 64 lanes; values for lane l, diagonal j at
// cfs_plan::packet_val_pos<V>(l, j), slots at l*4+j
__device__ __forceinline__ void load_packet(const double *tv, uint32_t off, int lane,
                                            double (&v)[4]) {
  const double2 lo = *reinterpret_cast<const double2 *>(tv + off + lane * 2);
  const double2 hi = *reinterpret_cast<const double2 *>(tv + off + 128 + lane * 2);
  v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
}
__device__ __forceinline__ void load_packet(const float *tv, uint32_t off, int lane,
                                            float (&v)[4]) {
  const float4 q = *reinterpret_cast<const float4 *>(tv + off + lane * 4);
  v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}

// ---------------------------------------------------------------------------
// tile kernel: persistent workgroups, each walks its group of tiles.
//   prologue: x window -> LDS (own rows coalesced, halo gathered), y window = 0
//   slices  : one lane = one row; row-side sum in a register, transposed
//             updates into the LDS y window with ds_add_f64 / ds_add_f32
//   epilogue: own rows -> y (plain coalesced stores, fully overwrites y),
//             halo sums -> this tile's private strip (plain coalesced stores)
// ---------------------------------------------------------------------------
template <typename V, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
    cfs_sym_tile_kernel(const SymDev<V> d, const V *__restrict__ x, V *__restrict__ y) {
  extern __shared__ __align__(16) unsigned char cfs_smem[];
  V *xl = reinterpret_cast<V *>(cfs_smem);
  V *yl = xl + d.lds_slots;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = BLOCK / 64;
  // blocks b and b+8 share an XCD (round-robin dispatch): give every XCD a
  // contiguous run of groups so neighbouring tiles share halo lines in one L2.
  // Placement only affects speed, never correctness.
  const int nper = gridDim.x >> 3;
  const int g = (blockIdx.x & 7) * nper + (blockIdx.x >> 3);
  const int t0 = d.group_ptr[g], t1 = d.group_ptr[g + 1];

  for (int ti = t0; ti < t1; ++ti) {
    const Tile t = d.tiles[ti];
    const int nown = t.nown, nslots = t.nslots;
    const int lrow0 = t.row0 - d.row_begin;

    for (int s = tid; s < nown; s += BLOCK) {
      xl[s] = x[t.row0 + s];
      yl[s] = V(0);
    }
    for (int s = nown + tid; s < nslots; s += BLOCK) {
      xl[s] = x[d.halo_col[t.halo_off + (s - nown)]];
      yl[s] = V(0);
    }
    __syncthreads();

    const V *tv = d.vals + t.nnz_off;
    const uint16_t *ts = d.slots + t.nnz_off;
    for (int s = wave; s < t.nslices; s += NW) {
      const int p = s * 64 + lane;
      const bool has = p < nown;
      const uint32_t info = has ? d.rowinfo[lrow0 + p] : 0u;
      const int r = info & 0xffffu;
      const int len = (int)(info >> 16);
      const V dg = has ? d.diag[lrow0 + p] : V(0);
      const V xi = xl[r];
      V acc = V(0);
      uint32_t off = d.slice_off[t.slice_base + s];
      const int minlen = __builtin_amdgcn_readlane(len, 63);
      const int maxlen = __builtin_amdgcn_readfirstlane(len);
      const int nfull = minlen >> 2;

      for (int q = 0; q < nfull; ++q) {
        V v[4];
        load_packet(tv, off, lane, v);
        const ushort4 c = *reinterpret_cast<const ushort4 *>(ts + off + lane * 4);
        const unsigned cs[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc = fma(v[j], xl[cs[j]], acc);
          atomicAdd(&yl[cs[j]], v[j] * xi);
        }
        off += 256;
      }
      for (int k = nfull << 2; k < maxlen; ++k) {
        const bool act = k < len;
        const unsigned long long m = __ballot(act);
        if (act) {
          const V v = tv[off + lane];
          const unsigned c = ts[off + lane];
          acc = fma(v, xl[c], acc);
          atomicAdd(&yl[c], v * xi);
        }
        off += __popcll(m);
      }
      if (has) atomicAdd(&yl[r], fma(dg, xi, acc));
    }
    __syncthreads();

    for (int s = tid; s < nown; s += BLOCK) y[lrow0 + s] = yl[s];
    for (int s = nown + tid; s < nslots; s += BLOCK)
      d.strip[t.halo_off + (s - nown)] = yl[s];
    __syncthreads();
  }
}

// halo fold: y[dst] += sum of the strip entries aimed at dst, in fixed order
template <typename V>
__global__ void __launch_bounds__(256)
    cfs_fold_kernel(V *__restrict__ y, const V *__restrict__ src,
                    const int32_t *__restrict__ frow, const int32_t *__restrict__ fptr,
                    const int32_t *__restrict__ fidx, int m) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const int r = frow[i];
  V s = y[r];
  for (int q = fptr[i]; q < fptr[i + 1]; ++q) s += src[fidx[q]];
  y[r] = s;
}

// pack contributions for rows owned by lower ranks: one value per remote row
template <typename V>
__global__ void __launch_bounds__(256)
    cfs_pack_kernel(V *__restrict__ send, const V *__restrict__ src,
                    const int32_t *__restrict__ sptr, const int32_t *__restrict__ sidx,
                    int m) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  V s = V(0);
  for (int q = sptr[i]; q < sptr[i + 1]; ++q) s += src[sidx[q]];
  send[i] = s;
}

// general CSR: LPR lanes per row, shuffle reduction inside the sub-group
template <typename V, int LPR>
__global__ void __launch_bounds__(256)
    cfs_csr_kernel(int nrows, const int32_t *__restrict__ rowptr,
                   const int32_t *__restrict__ colind, const V *__restrict__ values,
                   const V *__restrict__ x, V *__restrict__ y) {
  const int gtid = blockIdx.x * 256 + threadIdx.x;
  const int row = gtid / LPR, sub = gtid % LPR;
  V acc = V(0);
  if (row < nrows) {
    const int b = rowptr[row], e = rowptr[row + 1];
    for (int j = b + sub; j < e; j += LPR) acc = fma(values[j], x[colind[j]], acc);
  }
#pragma unroll
  for (int o = LPR >> 1; o > 0; o >>= 1) acc += __shfl_down(acc, o, LPR);
  if (row < nrows && sub == 0) y[row] = acc;
}

// ---------------------------------------------------------------------------
// host objects
// ---------------------------------------------------------------------------
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int upload(const void *src, size_t n) {
    bytes = n;
    if (n == 0) n = 16;
    HIPCHK(hipMalloc(&p, n));
    if (bytes) HIPCHK(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    return 0;
  }
  int alloc(size_t n) {
    bytes = n;
    if (n == 0) n = 16;
    HIPCHK(hipMalloc(&p, n));
    return 0;
  }
};

struct cfs_hip_sym_s {
  int value_bytes = 8;
  virtual ~cfs_hip_sym_s() {}
  virtual int spmv_local(void *y, const void *x, void *send, hipStream_t st) = 0;
  virtual int recv_fold(void *y, const void *recv, hipStream_t st) = 0;
  virtual int set_recv(int nrecv, const int *rows) = 0;
  virtual void stats(cfs_hip_sym_stats *o) = 0;
  virtual const std::vector<int32_t> &send_counts() = 0;
  virtual const std::vector<int32_t> &send_rows() = 0;
  virtual int n() = 0;
  virtual int rows() = 0;
  // staging for host-pointer callers
  DevBuf xstage, ystage;
};

template <typename V> struct SymMatrix : cfs_hip_sym_s {
  SymPlan<V> P; // big arrays released after upload
  DevBuf tiles, group_ptr, halo_col, rowinfo, diag, slice_off, vals, slots, strip;
  DevBuf fold_row, fold_ptr, fold_idx, send_ptr, send_idx;
  DevBuf rfold_row, rfold_ptr, rfold_idx;
  SymDev<V> dev{};
  int nfold = 0, nsend = 0, nrfold = 0;
  size_t lds_bytes = 0;
  int64_t halo_slots = 0, stream_len = 0, nslices = 0;

  int upload() {
    int rc;
#define UP(buf, vec)                                                          \
  if ((rc = buf.upload(vec.data(), vec.size() * sizeof(vec[0])))) return rc;
    UP(tiles, P.tiles)
    UP(group_ptr, P.group_ptr)
    UP(halo_col, P.halo_col)
    UP(rowinfo, P.rowinfo)
    UP(diag, P.diag)
    UP(slice_off, P.slice_off)
    UP(vals, P.vals)
    UP(slots, P.slots)
    UP(fold_row, P.fold_row)
    UP(fold_ptr, P.fold_ptr)
    UP(fold_idx, P.fold_idx)
    UP(send_ptr, P.send_ptr)
    UP(send_idx, P.send_idx)
#undef UP
    if ((rc = strip.alloc(P.halo_col.size() * sizeof(V)))) return rc;
    halo_slots = (int64_t)P.halo_col.size();
    stream_len = P.stream_len;
    nslices = (int64_t)P.slice_off.size();
    nfold = (int)P.fold_row.size();
    nsend = (int)P.send_row.size();
    dev.tiles = (const Tile *)tiles.p;
    dev.group_ptr = (const int32_t *)group_ptr.p;
    dev.halo_col = (const int32_t *)halo_col.p;
    dev.rowinfo = (const uint32_t *)rowinfo.p;
    dev.diag = (const V *)diag.p;
    dev.slice_off = (const uint32_t *)slice_off.p;
    dev.vals = (const V *)vals.p;
    dev.slots = (const uint16_t *)slots.p;
    dev.strip = (V *)strip.p;
    dev.row_begin = P.row_begin;
    dev.lds_slots = P.lds_slots;
    lds_bytes = (size_t)P.lds_slots * 2 * sizeof(V);
    // release the big host arrays; keep the small metadata
    std::vector<V>().swap(P.vals);
    std::vector<uint16_t>().swap(P.slots);
    std::vector<V>().swap(P.diag);
    std::vector<uint32_t>().swap(P.rowinfo);
    std::vector<int32_t>().swap(P.fold_idx);
    std::vector<int32_t>().swap(P.send_idx);
    return raise_lds_limit();
  }

  template <int BLOCK> int raise_one() {
    HIPCHK(hipFuncSetAttribute((const void *)cfs_sym_tile_kernel<V, BLOCK>,
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds_bytes));
    return 0;
  }
  int raise_lds_limit() {
    switch (P.block_threads) {
    case 256: return raise_one<256>();
    case 512: return raise_one<512>();
    default: return raise_one<1024>();
    }
  }

  template <int BLOCK> void launch_tiles(V *y, const V *x, hipStream_t st) {
    hipLaunchKernelGGL((cfs_sym_tile_kernel<V, BLOCK>), dim3(P.ngroups), dim3(BLOCK),
                       lds_bytes, st, dev, x, y);
  }

  int spmv_local(void *yv, const void *xv, void *sendv, hipStream_t st) override {
    V *y = (V *)yv;
    const V *x = (const V *)xv;
    if (P.tiles.empty()) return 0;
    switch (P.block_threads) {
    case 256: launch_tiles<256>(y, x, st); break;
    case 512: launch_tiles<512>(y, x, st); break;
    default: launch_tiles<1024>(y, x, st); break;
    }
    if (nfold > 0)
      hipLaunchKernelGGL((cfs_fold_kernel<V>), dim3((nfold + 255) / 256), dim3(256), 0,
                         st, y, (const V *)strip.p, (const int32_t *)fold_row.p,
                         (const int32_t *)fold_ptr.p, (const int32_t *)fold_idx.p, nfold);
    if (nsend > 0) {
      if (!sendv) return set_err(CFS_HIP_ERR_ARG, "shard has remote rows: send_buf required");
      hipLaunchKernelGGL((cfs_pack_kernel<V>), dim3((nsend + 255) / 256), dim3(256), 0,
                         st, (V *)sendv, (const V *)strip.p, (const int32_t *)send_ptr.p,
                         (const int32_t *)send_idx.p, nsend);
    }
    HIPCHK(hipGetLastError());
    return 0;
  }

  int set_recv(int nrecv, const int *rows) override {
    if (!cfs_plan::set_recv(P, nrecv, rows)) return set_err(CFS_HIP_ERR_ARG, P.error);
    int rc;
    rfold_row = DevBuf();
    rfold_ptr = DevBuf();
    rfold_idx = DevBuf();
    if ((rc = rfold_row.upload(P.rfold_row.data(), P.rfold_row.size() * 4))) return rc;
    if ((rc = rfold_ptr.upload(P.rfold_ptr.data(), P.rfold_ptr.size() * 4))) return rc;
    if ((rc = rfold_idx.upload(P.rfold_idx.data(), P.rfold_idx.size() * 4))) return rc;
    nrfold = (int)P.rfold_row.size();
    return 0;
  }

  int recv_fold(void *yv, const void *recv, hipStream_t st) override {
    if (nrfold > 0)
      hipLaunchKernelGGL((cfs_fold_kernel<V>), dim3((nrfold + 255) / 256), dim3(256), 0,
                         st, (V *)yv, (const V *)recv, (const int32_t *)rfold_row.p,
                         (const int32_t *)rfold_ptr.p, (const int32_t *)rfold_idx.p, nrfold);
    HIPCHK(hipGetLastError());
    return 0;
  }

  void stats(cfs_hip_sym_stats *o) override {
    memset(o, 0, sizeof *o);
    const int64_t s = sizeof(V), rows_ = P.row_end - P.row_begin;
    o->n = P.n;
    o->row_begin = P.row_begin;
    o->row_end = P.row_end;
    o->value_bytes = (int)s;
    o->nnz_low = P.nnz_low;
    o->nnz_diag = P.nnz_diag;
    o->nnz_full = P.nnz_full;
    o->ntiles = (int)P.tiles.size();
    o->nslices = (int)nslices;
    o->max_slots_used = P.lds_slots;
    o->block_threads = P.block_threads;
    o->halo_slots = halo_slots;
    o->fold_rows = nfold;
    o->remote_vals = nsend;
    o->lds_bytes = (int64_t)lds_bytes;
    o->bytes_algorithmic = P.nnz_low * (4 + s) + rows_ * (4 + 3 * s);
    o->bytes_streamed = stream_len * (s + 2) + rows_ * (4 + 3 * s) +
                        halo_slots * (4 + 2 * s)            /* halo_col, x, strip st */
                        + halo_slots * (4 + s)              /* fold: idx + strip ld  */
                        + (int64_t)(nfold + nsend) * (8 + 2 * s) + nslices * 4 +
                        (int64_t)P.tiles.size() * (int64_t)sizeof(Tile);
    o->device_bytes = (int64_t)(tiles.bytes + group_ptr.bytes + halo_col.bytes +
                                rowinfo.bytes + diag.bytes + slice_off.bytes + vals.bytes +
                                slots.bytes + strip.bytes + fold_row.bytes + fold_ptr.bytes +
                                fold_idx.bytes + send_ptr.bytes + send_idx.bytes);
  }
  const std::vector<int32_t> &send_counts() override { return P.send_counts; }
  const std::vector<int32_t> &send_rows() override { return P.send_row; }
  int n() override { return P.n; }
  int rows() override { return P.row_end - P.row_begin; }
};

struct cfs_hip_csr_s {
  int value_bytes = 8, nrows = 0, ncols = 0, lpr = 16;
  int64_t nnz = 0;
  DevBuf rowptr, colind, values, xstage, ystage;
};

// ---------------------------------------------------------------------------
// C ABI (every function below is declared extern "C" in cfs_hip.h)
// ---------------------------------------------------------------------------

int cfs_hip_abi_version(void) { return CFS_HIP_ABI_VERSION; }
const char *cfs_hip_last_error(void) { return g_err.c_str(); }

int cfs_hip_device_count(int *count) {
  if (!count) return set_err(CFS_HIP_ERR_ARG, "count is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *count = 0;
    return set_err(CFS_HIP_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = c;
  return 0;
}

int cfs_hip_init(int device) {
  if (g_device == device && g_stream) return 0;
  HIPCHK(hipSetDevice(device));
  if (g_stream) {
    (void)hipStreamDestroy(g_stream);
    g_stream = nullptr;
  }
  HIPCHK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
  g_device = device;
  return 0;
}

int cfs_hip_default_stream(void **stream) {
  int rc = ensure_init();
  if (rc) return rc;
  *stream = (void *)g_stream;
  return 0;
}

int cfs_hip_synchronize(void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int cfs_hip_alloc(size_t bytes, int kind, void **out) {
  if (!out) return set_err(CFS_HIP_ERR_ARG, "out is NULL");
  int rc = ensure_init();
  if (rc) return rc;
  if (bytes == 0) bytes = 64;
  if (kind == CFS_HIP_MEM_DEVICE) HIPCHK(hipMalloc(out, bytes));
  else if (kind == CFS_HIP_MEM_PINNED) HIPCHK(hipHostMalloc(out, bytes, hipHostMallocDefault));
  else return set_err(CFS_HIP_ERR_ARG, "unknown memory kind");
  return 0;
}

int cfs_hip_free(void *p, int kind) {
  if (!p) return 0;
  if (kind == CFS_HIP_MEM_DEVICE) HIPCHK(hipFree(p));
  else if (kind == CFS_HIP_MEM_PINNED) HIPCHK(hipHostFree(p));
  else return set_err(CFS_HIP_ERR_ARG, "unknown memory kind");
  return 0;
}

int cfs_hip_memcpy(void *dst, const void *src, size_t bytes, int dir) {
  int rc = ensure_init();
  if (rc) return rc;
  hipMemcpyKind k = dir == CFS_HIP_H2D   ? hipMemcpyHostToDevice
                    : dir == CFS_HIP_D2H ? hipMemcpyDeviceToHost
                                         : hipMemcpyDeviceToDevice;
  HIPCHK(hipMemcpy(dst, src, bytes, k));
  return 0;
}

int cfs_hip_memset(void *dst, int value, size_t bytes) {
  HIPCHK(hipMemset(dst, value, bytes));
  return 0;
}

static cfs_plan::Options to_opts(const cfs_hip_options *o) {
  cfs_plan::Options r;
  if (o) {
    r.max_slots = o->max_slots;
    r.max_tile_nnz = o->max_tile_nnz;
    r.block_threads = o->block_threads;
    r.flags = o->flags;
  }
  return r;
}

template <typename V>
static int sym_create(int n, const int *rowptr, const int *colind, const V *values,
                      int nranks, int rank, const int *row_splits,
                      const cfs_hip_options *opt, cfs_hip_sym_t *out) {
  if (!out) return set_err(CFS_HIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (n < 0 || !rowptr || (n > 0 && rowptr[n] > 0 && (!colind || !values)))
    return set_err(CFS_HIP_ERR_ARG, "null CSR array");
  if (nranks < 1 || rank < 0 || rank >= nranks)
    return set_err(CFS_HIP_ERR_ARG, "bad rank / nranks");
  if (nranks > 1 && !row_splits) return set_err(CFS_HIP_ERR_ARG, "row_splits required");
  int rc = ensure_init();
  if (rc) return rc;
  auto *m = new SymMatrix<V>();
  m->value_bytes = (int)sizeof(V);
  if (!cfs_plan::build_plan<V>(n, rowptr, colind, values, nranks, rank,
                               nranks > 1 ? row_splits : nullptr, to_opts(opt), m->P)) {
    std::string e = m->P.error;
    delete m;
    return set_err(CFS_HIP_ERR_UNSUPPORTED, e);
  }
  rc = m->upload();
  if (rc) {
    delete m;
    return rc;
  }
  *out = m;
  return 0;
}

int cfs_hip_sym_create_f64(int n, const int *rowptr, const int *colind, const double *values,
                           const cfs_hip_options *opt, cfs_hip_sym_t *out) {
  return sym_create<double>(n, rowptr, colind, values, 1, 0, nullptr, opt, out);
}
int cfs_hip_sym_create_f32(int n, const int *rowptr, const int *colind, const float *values,
                           const cfs_hip_options *opt, cfs_hip_sym_t *out) {
  return sym_create<float>(n, rowptr, colind, values, 1, 0, nullptr, opt, out);
}
int cfs_hip_sym_create_shard_f64(int n, const int *rowptr, const int *colind,
                                 const double *values, int nranks, int rank,
                                 const int *row_splits, const cfs_hip_options *opt,
                                 cfs_hip_sym_t *out) {
  return sym_create<double>(n, rowptr, colind, values, nranks, rank, row_splits, opt, out);
}
int cfs_hip_sym_create_shard_f32(int n, const int *rowptr, const int *colind,
                                 const float *values, int nranks, int rank,
                                 const int *row_splits, const cfs_hip_options *opt,
                                 cfs_hip_sym_t *out) {
  return sym_create<float>(n, rowptr, colind, values, nranks, rank, row_splits, opt, out);
}

int cfs_hip_sym_balanced_splits(int n, const int *rowptr, const int *colind, int nranks,
                                int *row_splits) {
  if (n < 0 || !rowptr || !row_splits || nranks < 1)
    return set_err(CFS_HIP_ERR_ARG, "bad argument");
  cfs_plan::balanced_splits(n, rowptr, colind, nranks, row_splits);
  return 0;
}

int cfs_hip_sym_destroy(cfs_hip_sym_t h) {
  delete h;
  return 0;
}

static bool is_device_ptr(const void *p) {
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError(); // unregistered host memory: clear the sticky error
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

int cfs_hip_sym_spmv_async(cfs_hip_sym_t h, void *y, const void *x, void *stream) {
  if (!h || !y || !x) return set_err(CFS_HIP_ERR_ARG, "null argument");
  if (!h->send_rows().empty())
    return set_err(CFS_HIP_ERR_ARG, "sharded handle: use cfs_hip_sym_spmv_local_async");
  return h->spmv_local(y, x, nullptr, (hipStream_t)stream);
}

int cfs_hip_sym_spmv(cfs_hip_sym_t h, void *y, const void *x) {
  if (!h || !y || !x) return set_err(CFS_HIP_ERR_ARG, "null argument");
  const size_t vb = (size_t)h->value_bytes;
  const bool xd = is_device_ptr(x), yd = is_device_ptr(y);
  const void *xdev = x;
  void *ydev = y;
  int rc;
  if (!xd) { // slow staged path for drop-in host-pointer callers
    if (!h->xstage.p && (rc = h->xstage.alloc((size_t)h->n() * vb))) return rc;
    HIPCHK(hipMemcpyAsync(h->xstage.p, x, (size_t)h->n() * vb, hipMemcpyHostToDevice, g_stream));
    xdev = h->xstage.p;
  }
  if (!yd) {
    if (!h->ystage.p && (rc = h->ystage.alloc((size_t)h->rows() * vb))) return rc;
    ydev = h->ystage.p;
  }
  if ((rc = cfs_hip_sym_spmv_async(h, ydev, xdev, g_stream))) return rc;
  if (!yd)
    HIPCHK(hipMemcpyAsync(y, ydev, (size_t)h->rows() * vb, hipMemcpyDeviceToHost, g_stream));
  HIPCHK(hipStreamSynchronize(g_stream));
  return 0;
}

int cfs_hip_sym_shard_send_counts(cfs_hip_sym_t h, int *send_counts) {
  if (!h || !send_counts) return set_err(CFS_HIP_ERR_ARG, "null argument");
  const auto &c = h->send_counts();
  for (size_t i = 0; i < c.size(); i++) send_counts[i] = c[i];
  return 0;
}
int cfs_hip_sym_shard_send_rows(cfs_hip_sym_t h, int *rows) {
  if (!h || !rows) return set_err(CFS_HIP_ERR_ARG, "null argument");
  const auto &r = h->send_rows();
  for (size_t i = 0; i < r.size(); i++) rows[i] = r[i];
  return 0;
}
int cfs_hip_sym_shard_set_recv(cfs_hip_sym_t h, int nrecv, const int *recv_rows) {
  if (!h || (nrecv > 0 && !recv_rows)) return set_err(CFS_HIP_ERR_ARG, "null argument");
  return h->set_recv(nrecv, recv_rows);
}
int cfs_hip_sym_spmv_local_async(cfs_hip_sym_t h, void *y, const void *x, void *send,
                                 void *stream) {
  if (!h || !y || !x) return set_err(CFS_HIP_ERR_ARG, "null argument");
  return h->spmv_local(y, x, send, (hipStream_t)stream);
}
int cfs_hip_sym_recv_fold_async(cfs_hip_sym_t h, void *y, const void *recv, void *stream) {
  if (!h || !y) return set_err(CFS_HIP_ERR_ARG, "null argument");
  return h->recv_fold(y, recv, (hipStream_t)stream);
}

int cfs_hip_sym_get_stats(cfs_hip_sym_t h, cfs_hip_sym_stats *out) {
  if (!h || !out) return set_err(CFS_HIP_ERR_ARG, "null argument");
  h->stats(out);
  return 0;
}

// ---- host-only plan self-check (no device needed) -------------------------
template <typename V>
static int plan_check(int n, const int *rowptr, const int *colind, const V *values,
                      int nranks, int rank, const int *row_splits,
                      const cfs_hip_options *opt, cfs_hip_plan_report *rep) {
  if (!rep) return set_err(CFS_HIP_ERR_ARG, "report is NULL");
  memset(rep, 0, sizeof *rep);
  SymPlan<V> P;
  if (!cfs_plan::build_plan<V>(n, rowptr, colind, values, nranks, rank,
                               nranks > 1 ? row_splits : nullptr, to_opts(opt), P))
    return set_err(CFS_HIP_ERR_UNSUPPORTED, P.error);
  std::vector<int32_t> r, c;
  std::vector<V> v;
  cfs_plan::decode_plan(P, r, c, v);
  rep->ntiles = (int)P.tiles.size();
  rep->ngroups = P.ngroups;
  rep->nslices = (int64_t)P.slice_off.size();
  rep->halo_slots = (int64_t)P.halo_col.size();
  rep->stream_len = P.stream_len;
  rep->nnz_low = P.nnz_low;
  rep->lds_slots = P.lds_slots;
  rep->fold_rows = (int64_t)P.fold_row.size();
  rep->remote_vals = (int64_t)P.send_row.size();
  rep->decoded = (int64_t)r.size();
  // (1) decoded triples == strict lower triangle of the owned rows (per-row
  // multisets; order inside a row is preserved by construction)
  int64_t bad = 0;
  {
    std::vector<int64_t> pos(P.row_end - P.row_begin + 1, 0);
    for (size_t k = 0; k < r.size(); k++) pos[r[k] - P.row_begin + 1]++;
    for (size_t i = 1; i < pos.size(); i++) pos[i] += pos[i - 1];
    std::vector<int32_t> dc(r.size());
    std::vector<V> dv(r.size());
    std::vector<int64_t> fill(pos.begin(), pos.end() - 1);
    for (size_t k = 0; k < r.size(); k++) {
      int64_t q = fill[r[k] - P.row_begin]++;
      dc[q] = c[k];
      dv[q] = v[k];
    }
    for (int i = P.row_begin; i < P.row_end; i++) {
      int64_t q = pos[i - P.row_begin];
      for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
        if (colind[j] >= i) continue;
        if (q >= pos[i - P.row_begin + 1] || dc[q] != colind[j] ||
            memcmp(&dv[q], &values[j], sizeof(V)) != 0)
          bad++;
        q++;
      }
      if (q != pos[i - P.row_begin + 1]) bad++;
    }
  }
  // (2) fold + send indices cover every strip entry exactly once and point at
  // a strip entry whose column is the destination row
  {
    std::vector<char> seen(P.halo_col.size(), 0);
    for (size_t i = 0; i < P.fold_row.size(); i++)
      for (int q = P.fold_ptr[i]; q < P.fold_ptr[i + 1]; q++) {
        int s = P.fold_idx[q];
        if (seen[s]++ || P.halo_col[s] != P.fold_row[i] + P.row_begin) bad++;
      }
    for (size_t i = 0; i < P.send_row.size(); i++)
      for (int q = P.send_ptr[i]; q < P.send_ptr[i + 1]; q++) {
        int s = P.send_idx[q];
        if (seen[s]++ || P.halo_col[s] != P.send_row[i]) bad++;
      }
    for (char s : seen)
      if (!s) bad++;
  }
  // (3) groups partition the tiles; tiles partition the rows
  {
    if (P.group_ptr.front() != 0 || P.group_ptr.back() != (int)P.tiles.size()) bad++;
    for (size_t g = 1; g < P.group_ptr.size(); g++)
      if (P.group_ptr[g] < P.group_ptr[g - 1]) bad++;
    int row = P.row_begin;
    for (const Tile &t : P.tiles) {
      if (t.row0 != row || t.nslots > P.max_slots || t.nslots > P.lds_slots) bad++;
      row += t.nown;
    }
    if (row != P.row_end) bad++;
  }
  rep->mismatches = bad;
  return 0;
}

int cfs_hip_sym_plan_check_f64(int n, const int *rowptr, const int *colind,
                               const double *values, int nranks, int rank,
                               const int *row_splits, const cfs_hip_options *opt,
                               cfs_hip_plan_report *rep) {
  return plan_check<double>(n, rowptr, colind, values, nranks, rank, row_splits, opt, rep);
}
int cfs_hip_sym_plan_check_f32(int n, const int *rowptr, const int *colind,
                               const float *values, int nranks, int rank,
                               const int *row_splits, const cfs_hip_options *opt,
                               cfs_hip_plan_report *rep) {
  return plan_check<float>(n, rowptr, colind, values, nranks, rank, row_splits, opt, rep);
}

// ---- general CSR ------------------------------------------------------------
template <typename V>
static int csr_create(int nrows, int ncols, const int *rowptr, const int *colind,
                      const V *values, cfs_hip_csr_t *out) {
  if (!out || !rowptr || nrows < 0) return set_err(CFS_HIP_ERR_ARG, "bad argument");
  int rc = ensure_init();
  if (rc) return rc;
  auto *m = new cfs_hip_csr_s();
  m->value_bytes = (int)sizeof(V);
  m->nrows = nrows;
  m->ncols = ncols;
  m->nnz = rowptr[nrows];
  if ((rc = m->rowptr.upload(rowptr, ((size_t)nrows + 1) * 4)) ||
      (rc = m->colind.upload(colind, (size_t)m->nnz * 4)) ||
      (rc = m->values.upload(values, (size_t)m->nnz * sizeof(V)))) {
    delete m;
    return rc;
  }
  double avg = nrows ? (double)m->nnz / nrows : 0;
  m->lpr = avg > 48 ? 64 : avg > 24 ? 32 : avg > 12 ? 16 : avg > 6 ? 8 : 4;
  *out = m;
  return 0;
}
int cfs_hip_csr_create_f64(int nrows, int ncols, const int *rowptr, const int *colind,
                           const double *values, cfs_hip_csr_t *out) {
  return csr_create<double>(nrows, ncols, rowptr, colind, values, out);
}
int cfs_hip_csr_create_f32(int nrows, int ncols, const int *rowptr, const int *colind,
                           const float *values, cfs_hip_csr_t *out) {
  return csr_create<float>(nrows, ncols, rowptr, colind, values, out);
}

template <typename V, int LPR>
static void csr_launch(cfs_hip_csr_t h, V *y, const V *x, hipStream_t st) {
  long threads = (long)h->nrows * LPR;
  int blocks = (int)((threads + 255) / 256);
  if (blocks == 0) return;
  hipLaunchKernelGGL((cfs_csr_kernel<V, LPR>), dim3(blocks), dim3(256), 0, st, h->nrows,
                     (const int32_t *)h->rowptr.p, (const int32_t *)h->colind.p,
                     (const V *)h->values.p, x, y);
}
template <typename V> static void csr_dispatch(cfs_hip_csr_t h, V *y, const V *x, hipStream_t st) {
  switch (h->lpr) {
  case 64: csr_launch<V, 64>(h, y, x, st); break;
  case 32: csr_launch<V, 32>(h, y, x, st); break;
  case 16: csr_launch<V, 16>(h, y, x, st); break;
  case 8: csr_launch<V, 8>(h, y, x, st); break;
  default: csr_launch<V, 4>(h, y, x, st); break;
  }
}

int cfs_hip_csr_spmv_async(cfs_hip_csr_t h, void *y, const void *x, void *stream) {
  if (!h || !y || !x) return set_err(CFS_HIP_ERR_ARG, "null argument");
  hipStream_t st = (hipStream_t)stream;
  if (h->value_bytes == 8) csr_dispatch<double>(h, (double *)y, (const double *)x, st);
  else csr_dispatch<float>(h, (float *)y, (const float *)x, st);
  HIPCHK(hipGetLastError());
  return 0;
}

int cfs_hip_csr_spmv(cfs_hip_csr_t h, void *y, const void *x) {
  if (!h || !y || !x) return set_err(CFS_HIP_ERR_ARG, "null argument");
  const size_t vb = (size_t)h->value_bytes;
  const bool xd = is_device_ptr(x), yd = is_device_ptr(y);
  const void *xdev = x;
  void *ydev = y;
  int rc;
  if (!xd) {
    if (!h->xstage.p && (rc = h->xstage.alloc((size_t)h->ncols * vb))) return rc;
    HIPCHK(hipMemcpyAsync(h->xstage.p, x, (size_t)h->ncols * vb, hipMemcpyHostToDevice, g_stream));
    xdev = h->xstage.p;
  }
  if (!yd) {
    if (!h->ystage.p && (rc = h->ystage.alloc((size_t)h->nrows * vb))) return rc;
    ydev = h->ystage.p;
  }
  if ((rc = cfs_hip_csr_spmv_async(h, ydev, xdev, g_stream))) return rc;
  if (!yd)
    HIPCHK(hipMemcpyAsync(y, ydev, (size_t)h->nrows * vb, hipMemcpyDeviceToHost, g_stream));
  HIPCHK(hipStreamSynchronize(g_stream));
  return 0;
}

int cfs_hip_csr_destroy(cfs_hip_csr_t h) {
  delete h;
  return 0;
}

// ---- events -------------------------------------------------------------------
int cfs_hip_event_create(void **ev) {
  int rc = ensure_init();
  if (rc) return rc;
  hipEvent_t e;
  HIPCHK(hipEventCreate(&e));
  *ev = (void *)e;
  return 0;
}
int cfs_hip_event_record(void *ev, void *stream) {
  HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return 0;
}
int cfs_hip_event_elapsed_ms(void *start, void *stop, float *ms) {
  HIPCHK(hipEventSynchronize((hipEvent_t)stop));
  HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return 0;
}
int cfs_hip_event_destroy(void *ev) {
  HIPCHK(hipEventDestroy((hipEvent_t)ev));
  return 0;
}
