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

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "cfs_hip.h"
#include "cfs_plan.hpp"
#include "cfs_runtime.hpp"
#include "cfs_comm.hpp"

using cfs_plan::SymPlan;
using cfs_plan::Tile;

// ---------------------------------------------------------------------------
// runtime (cfs_runtime.hpp): per-device contexts, pools, error plumbing
// ---------------------------------------------------------------------------
using cfs_rt::DevBuf;
using cfs_rt::DeviceGuard;
using cfs_rt::PinBuf;
using cfs_rt::set_err;

static int ensure_init() { return cfs_rt::ensure_home(); }

// copy with non-temporal stores (dst 16-byte aligned): a staging block is written once and read by the
// DMA engine, never by this CPU -- ordinary stores would first read every destination line into the cache
// (read-for-ownership: three memory transfers per byte instead of two; glibc only switches to streaming
// stores for copies far larger than one thread's 1 MiB share of a piece)
static void stream_copy(char *dst, const char *src, size_t n) {
  typedef long long v2 __attribute__((vector_size(16), aligned(16)));
  typedef long long v2u __attribute__((vector_size(16), aligned(1)));
  if (((uintptr_t)dst & 15) != 0) {
    memcpy(dst, src, n);
    return;
  }
  size_t i = 0;
  for (; i + 64 <= n; i += 64) {
    const v2u a = *reinterpret_cast<const v2u *>(src + i), b = *reinterpret_cast<const v2u *>(src + i + 16),
              c = *reinterpret_cast<const v2u *>(src + i + 32), d = *reinterpret_cast<const v2u *>(src + i + 48);
    __builtin_nontemporal_store((v2)a, reinterpret_cast<v2 *>(dst + i));
    __builtin_nontemporal_store((v2)b, reinterpret_cast<v2 *>(dst + i + 16));
    __builtin_nontemporal_store((v2)c, reinterpret_cast<v2 *>(dst + i + 32));
    __builtin_nontemporal_store((v2)d, reinterpret_cast<v2 *>(dst + i + 48));
  }
  if (i < n) memcpy(dst + i, src + i, n - i);
  __builtin_ia32_sfence();
}
void cfs_rt::parallel_copy(void *dst, const void *src, size_t bytes) {
  const int T = cfs_plan::host_threads();
  static const bool stream = !(getenv("CFS_HIP_STREAM_COPY") && atoi(getenv("CFS_HIP_STREAM_COPY")) == 0);
  if (bytes < ((size_t)1 << 20) || T < 2) {
    memcpy(dst, src, bytes);
    return;
  }
  const size_t chunk = ((bytes + T - 1) / T + 4095) & ~(size_t)4095;
#pragma omp parallel for schedule(static) num_threads(T)
  for (int t = 0; t < T; t++) {
    const size_t b = (size_t)t * chunk;
    if (b >= bytes) continue;
    const size_t len = std::min(chunk, bytes - b);
    if (stream) stream_copy((char *)dst + b, (const char *)src + b, len);
    else memcpy((char *)dst + b, (const char *)src + b, len);
  }
}

// ---------------------------------------------------------------------------
// device view of a plan
// ---------------------------------------------------------------------------
template <typename V> struct SymDev {
  const Tile *tiles;
  const Tile *gfirst;
  const int2 *group_range; // per launch slot: [first, last) tile
  const int32_t *slot_col; // original column of every slot of every tile
  const uint32_t *rowinfo;
  const V *diag;
  const uint4 *slice_meta; // {value offset, slot offset | lanes of packet 0 << 25, leader mask}
  const uint8_t *leadlane; // per slice and lane: the lane whose slots it reads
  const V *vals;
  const uint16_t *slots;
  const V *cvals; // COO leftovers (len % 4 per row): value, row slot, column slot
  const uint16_t *crows;
  const uint16_t *ccols;
  const V *fvals; // FAR sections (HYB): value, own row slot, global column
  const uint16_t *frows;
  const int32_t *fcols;
  V *strip;
  int row_begin;
  int lds_slots;
};

// one packet = 4 consecutive entries of the rows held by lanes [0, cnt): values
// for lane l, entry j at cfs_plan::packet_val_pos<V>(l, j, cnt), slots at l*4+j.
// Every lane issues the loads (lanes >= cnt re-read lane cnt-1's bytes: same
// cache lines, no extra DRAM traffic) so that the loads are unconditional
// instructions and the compiler's vmcnt bookkeeping stays exact.
template <typename V> struct Pkt {
  V v[4];
  ushort4 c;
};

// `leaders` (one bit per lane of the slice): a lane whose slot sequence is a
// prefix of an earlier lane's stores no slots and reads that LEADER lane's
// (cfs_plan::SymPlan::leadlane, one byte per lane).  A leader's block inside a
// packet's slot block is its rank among the leaders: lanes are sorted by packet
// count, so every leader below it is active whenever it is.
__device__ __forceinline__ int leader_rank(unsigned long long leaders, int leader_lane) {
  return __popcll(leaders & ((1ull << leader_lane) - 1ull));
}
// leaders among the cnt active lanes of a packet
__device__ __forceinline__ int active_leaders(unsigned long long leaders, int cnt) {
  return cnt > 0 ? __popcll(leaders & (~0ull >> (64 - cnt))) : 0;
}
// A matrix stream larger than the 256 MiB Infinity Cache is read exactly once
// per SpMV: its loads then carry the non-temporal hint (NT) so that they do not
// displace x, y, the slot tables and the strips from L2 / Infinity Cache
// (measured: Flan stand-in 0.117 -> 0.110 ms).  A stream that FITS the cache
// (ldoor, pwtk stand-ins) is served from it on every SpMV after the first and
// must stay cacheable (ldoor: 0.049 ms plain vs 0.057 ms with NT).
typedef double cfs_d2 __attribute__((ext_vector_type(2)));
typedef float cfs_f4 __attribute__((ext_vector_type(4)));
typedef unsigned short cfs_us4 __attribute__((ext_vector_type(4)));
template <bool NT, typename T> __device__ __forceinline__ T stream_load(const T *p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}
// lrank = rank of this lane's leader; a lane past the packet's end is clamped to
// the last stored block (same cache lines, bytes unused)
template <bool NT>
__device__ __forceinline__ void fetch_packet(Pkt<double> &p, const double *tv,
                                             const uint16_t *ts, uint32_t off, uint32_t soff,
                                             int cnt, unsigned long long leaders, int lane,
                                             int lrank) {
  const int ll = min(lane, max(cnt, 1) - 1);
  const int rk = min(lrank, max(active_leaders(leaders, cnt), 1) - 1);
  const cfs_d2 lo = stream_load<NT>(reinterpret_cast<const cfs_d2 *>(tv + off + ll * 2));
  const cfs_d2 hi = stream_load<NT>(reinterpret_cast<const cfs_d2 *>(tv + off + 2 * cnt + ll * 2));
  const cfs_us4 c = stream_load<NT>(reinterpret_cast<const cfs_us4 *>(ts + soff + rk * 4));
  p.c = make_ushort4(c.x, c.y, c.z, c.w);
  p.v[0] = lo.x; p.v[1] = lo.y; p.v[2] = hi.x; p.v[3] = hi.y;
}
template <bool NT>
__device__ __forceinline__ void fetch_packet(Pkt<float> &p, const float *tv,
                                             const uint16_t *ts, uint32_t off, uint32_t soff,
                                             int cnt, unsigned long long leaders, int lane,
                                             int lrank) {
  const int ll = min(lane, max(cnt, 1) - 1);
  const int rk = min(lrank, max(active_leaders(leaders, cnt), 1) - 1);
  const cfs_f4 q = stream_load<NT>(reinterpret_cast<const cfs_f4 *>(tv + off + ll * 4));
  const cfs_us4 c = stream_load<NT>(reinterpret_cast<const cfs_us4 *>(ts + soff + rk * 4));
  p.c = make_ushort4(c.x, c.y, c.z, c.w);
  p.v[0] = q.x; p.v[1] = q.y; p.v[2] = q.z; p.v[3] = q.w;
}
// entries of the packet's slot block = 4 x (leaders among its cnt lanes)
__device__ __forceinline__ uint32_t slot_block(unsigned long long leaders, int cnt) {
  return 4u * (uint32_t)active_leaders(leaders, cnt);
}

// one stored nonzero a = A[row][col(c)]: row side into the register
// accumulator, transposed side into the LDS y window (ds_add_f64 / ds_add_f32)
//
// The y window is ALWAYS fp64: on gfx950 ds_add_f64 runs at the rate of the
// matrix stream while ds_add_f32 is ~4x slower under this access pattern
// (measured: fp32 tile kernel 0.370 ms with ds_add_f32 vs 0.084 ms without the
// transposed atomics), so the single-precision build accumulates the window in
// double and rounds once when the window is flushed.
// OFFB (a mirrored shard, cfs_plan::Options::mirror_offblock): slots >= ny are
// off-block columns; their entries are one-sided -- row side only, the rank that
// owns the column computes the transposed side from its own copy of the entry.
// The y window of a tile.  Default: one fp64 word per slot, ds_add_f64 -- the order in
// which the waves' updates of a slot arrive varies from run to run, and so do the last
// bits of y.  DETERMINISTIC (CFS_HIP_FLAG_DETERMINISTIC): every contribution p is
// turned into a fixed-point number of 2 x 40 bits below a PER-SLOT scale 2^e and added
// with INTEGER atomics (hi and lo word of the slot): integer addition is associative,
// so the sums -- and y -- are bit-identical whatever the order.  e = (exponent bound of
// the 1-norm of the slot's matrix row, cfs_plan: 2^ex > sum_j |a_ij|) + (exponent of
// the largest |x| in the tile's window): a bound of EVERY partial sum of the slot, so
// the hi word never needs more than 40 bits, and a contribution keeps 2^-80 of its
// own row's scale -- the precision does not depend on how the matrix is scaled
// (round 2 used one scale per tile: rows 2^-15 below the tile's largest lost bits).
template <bool DET> struct YWin {
  double *y;         // !DET: fp64 sums.  DET: the hi words (as long long)
  long long *lo;     // DET: the lo words
  const short *ex;   // DET: per slot, exponent bound of the row's 1-norm (LDS)
  int xe;            // DET: exponent of the window's largest |x| (|x| < 2^xe), wave-uniform
  __device__ __forceinline__ void add(unsigned slot, double p) const {
    if (!DET) {
      atomicAdd(&y[slot], p);
    } else {
      const double q = ldexp(p, 40 - ((int)ex[slot] + xe)); // exact: a power of two
      const double h = trunc(q);
      const long long hi = (long long)h;
      const long long l2 = (long long)rint((q - h) * 0x1p40); // q - h is exact
      atomicAdd(reinterpret_cast<unsigned long long *>(y) + slot, (unsigned long long)hi);
      atomicAdd(reinterpret_cast<unsigned long long *>(lo) + slot, (unsigned long long)l2);
    }
  }
  __device__ __forceinline__ void zero(unsigned slot) const {
    if (!DET) {
      y[slot] = 0.0;
    } else {
      reinterpret_cast<long long *>(y)[slot] = 0;
      lo[slot] = 0;
    }
  }
  // value of a slot; mul = 1, or NaN when the window of x held a NaN / Inf
  __device__ __forceinline__ double get(unsigned slot, double mul) const {
    if (!DET) return y[slot];
    const long long hi = reinterpret_cast<const long long *>(y)[slot];
    const int e = (int)ex[slot];
    // (a row that holds a NaN / Inf has no fixed-point image either: it reads NaN)
    if (e >= cfs_plan::kExpNonFinite) return __longlong_as_double(0x7ff8000000000000ll);
    return ldexp((double)hi + (double)lo[slot] * 0x1p-40, e + xe - 40) * mul;
  }
};
// 2^k as a double, k clamped to the normal range
__device__ __forceinline__ double pow2_double(int k) {
  k = max(-1000, min(1000, k));
  return __longlong_as_double((long long)(k + 1023) << 52);
}

template <typename V, int MODE, bool OFFB, bool DET>
__device__ __forceinline__ void lds_update(const V *xl, const YWin<DET> &yw, V a, unsigned c, V xi,
                                           V &acc, unsigned ny) {
  if (MODE == 2) {
    acc = fma(a, xi + V(c), acc);
  } else {
    acc = fma(a, xl[c], acc);
    if (MODE == 0 && (!OFFB || c < ny)) yw.add(c, (double)a * (double)xi);
  }
}
template <typename V, int MODE, bool OFFB, bool DET>
__device__ __forceinline__ void consume_packet(const Pkt<V> &p, const V *xl, const YWin<DET> &yl,
                                               V xi, V &acc, unsigned ny) {
  lds_update<V, MODE, OFFB, DET>(xl, yl, p.v[0], p.c.x, xi, acc, ny);
  lds_update<V, MODE, OFFB, DET>(xl, yl, p.v[1], p.c.y, xi, acc, ny);
  lds_update<V, MODE, OFFB, DET>(xl, yl, p.v[2], p.c.z, xi, acc, ny);
  lds_update<V, MODE, OFFB, DET>(xl, yl, p.v[3], p.c.w, xi, acc, ny);
}
// The lanes of one mesh node read the same slots, i.e. their transposed updates of one
// packet entry go to the SAME y-window word: one instruction, up to three lanes on one
// address.  A follower that sits right behind a lane of its group hands its product to
// that lane (two DPP row shifts, runs of at most three lanes; cfs_plan sets the flags)
// and only the head of a run issues the atomic: fewer active lanes, no same-address
// serialisation.  `give` = my products go to my left neighbour, `take` = I add my right
// neighbour's.  Inactive lanes contribute 0 (a follower never has more packets than
// the lane to its left: lanes are sorted by packet count).
__device__ __forceinline__ double row_shl1(double v) { // lane l <- lane l + 1 (0 at the end of a row of 16)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x101, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x101, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <typename V, bool OFFB, bool DET>
__device__ __forceinline__ void entry_combined(const V *xl, const YWin<DET> &yw, V a, unsigned c, V xi,
                                               V &acc, unsigned ny, bool act, bool give, bool take) {
  double pr = 0.0;
  if (act) {
    acc = fma(a, xl[c], acc);
    pr = (double)a * (double)xi;
  }
  // (the shifts run with every lane enabled: a DPP read of a disabled lane returns nothing)
  const double t1 = row_shl1(pr);
  const double s1 = pr + (take ? t1 : 0.0);
  const double t2 = row_shl1(s1);
  const double q = pr + (take ? t2 : 0.0);
  if (act && !give && (!OFFB || c < ny)) yw.add(c, q);
}
template <typename V, bool OFFB, bool DET>
__device__ __forceinline__ void consume_combined(const Pkt<V> &p, const V *xl, const YWin<DET> &yl, V xi,
                                                 V &acc, unsigned ny, bool act, bool give, bool take) {
  entry_combined<V, OFFB, DET>(xl, yl, p.v[0], p.c.x, xi, acc, ny, act, give, take);
  entry_combined<V, OFFB, DET>(xl, yl, p.v[1], p.c.y, xi, acc, ny, act, give, take);
  entry_combined<V, OFFB, DET>(xl, yl, p.v[2], p.c.z, xi, acc, ny, act, give, take);
  entry_combined<V, OFFB, DET>(xl, yl, p.v[3], p.c.w, xi, acc, ny, act, give, take);
}
// one COO leftover a = A[row(r)][col(c)]: both sides through LDS atomics
template <typename V, int MODE, bool OFFB, bool DET>
__device__ __forceinline__ void coo_update(const V *xl, const YWin<DET> &yl, V a, unsigned r,
                                           unsigned c, unsigned ny) {
  if (MODE == 0 || MODE == 1) {
    yl.add(r, (double)a * (double)xl[c]);
    if (MODE == 0 && (!OFFB || c < ny)) yl.add(c, (double)a * (double)xl[r]);
  }
}

// ---------------------------------------------------------------------------
// tile kernel: persistent workgroups, each walks its group of tiles.
//   prologue: x window -> LDS (own rows coalesced, halo gathered), y window = 0
//   slices  : one lane = one row; row-side sum in a register, transposed
//             updates into the LDS y window with ds_add_f64 / ds_add_f32.
//             The matrix stream is software-pipelined per wave: the head
//             packet of the NEXT slice is requested a whole slice ahead and
//             packets ping-pong between two register sets, so a wave always
//             has loads in flight, also across slice boundaries.  The len%4
//             leftovers of the tile's rows are a flat COO section.
//   epilogue: own rows -> y (plain coalesced stores, fully overwrites y),
//             halo sums -> this tile's private strip (plain coalesced stores)
// MODE 0 is the product kernel.  MODE 1/2 are timing-only ablations selected by
// cfs_hip_options.flags (results are WRONG by construction; they exist to price
// the LDS atomics / LDS gathers against the pure matrix stream, never shipped
// as a result path): 1 = no transposed LDS atomics, 2 = no LDS traffic at all,
// 3 = windows only (no matrix stream), 4 = matrix stream only (no windows).
// ---------------------------------------------------------------------------
// (second launch bound: 4 waves per SIMD must stay resident -- two 512-thread
// workgroups, or one of 1 024, per CU -- i.e. at most 128 VGPRs; the persistent grid
// is sized for exactly that residency)
template <typename V, int BLOCK, int MODE, bool NT, bool OFFB, int U, bool DET = false, bool COMB = true>
__global__ void __launch_bounds__(BLOCK, 4)
    cfs_sym_tile_kernel(const Tile *__restrict__ a_tiles, const Tile *__restrict__ a_gfirst,
                        const int2 *__restrict__ a_group_range,
                        const int32_t *__restrict__ a_slot_col,
                        const uint32_t *__restrict__ a_rowinfo, const V *__restrict__ a_diag,
                        const uint4 *__restrict__ a_slice_meta,
                        const uint8_t *__restrict__ a_leadlane, const V *__restrict__ a_vals,
                        const uint16_t *__restrict__ a_slots, const V *__restrict__ a_cvals,
                        const uint16_t *__restrict__ a_crows,
                        const uint16_t *__restrict__ a_ccols, const V *__restrict__ a_fvals,
                        const uint16_t *__restrict__ a_frows, const int32_t *__restrict__ a_fcols,
                        V *__restrict__ a_strip, const int a_row_begin, const int a_lds_slots,
                        const V *__restrict__ x, V *__restrict__ y,
                        unsigned long long *__restrict__ dbg, const short *__restrict__ a_slot_exp) {
  // every array is a separate __restrict__ argument: read-only metadata at
  // wave-uniform addresses then becomes scalar loads (s_load), off the vector
  // memory counter the matrix stream is pipelined on
  struct {
    const Tile *__restrict__ tiles;
    const Tile *__restrict__ gfirst;
    const int2 *__restrict__ group_range;
    const int32_t *__restrict__ slot_col;
    const uint32_t *__restrict__ rowinfo;
    const V *__restrict__ diag;
    const uint4 *__restrict__ slice_meta;
    const uint8_t *__restrict__ leadlane;
    const V *__restrict__ vals;
    const uint16_t *__restrict__ slots;
    const V *__restrict__ cvals;
    const uint16_t *__restrict__ crows;
    const uint16_t *__restrict__ ccols;
    const V *__restrict__ fvals;
    const uint16_t *__restrict__ frows;
    const int32_t *__restrict__ fcols;
    V *__restrict__ strip;
    int row_begin, lds_slots;
  } d = {a_tiles, a_gfirst, a_group_range, a_slot_col, a_rowinfo, a_diag, a_slice_meta, a_leadlane, a_vals,
         a_slots, a_cvals, a_crows, a_ccols, a_fvals, a_frows, a_fcols, a_strip, a_row_begin, a_lds_slots};
  // slice ticket counter of the current tile (16 B so the dynamic region below
  // stays 16-byte aligned)
  __shared__ __align__(16) int cfs_ticket[4];
  extern __shared__ __align__(16) unsigned char cfs_smem[];
  // y window first (8-byte words: one per slot, two in the deterministic build), then x
  YWin<DET> yl;
  yl.y = reinterpret_cast<double *>(cfs_smem);
  yl.lo = reinterpret_cast<long long *>(cfs_smem) + (DET ? d.lds_slots : 0);
  V *xl = reinterpret_cast<V *>(yl.y + (DET ? 2 : 1) * d.lds_slots);
  short *exl = reinterpret_cast<short *>(xl + d.lds_slots); // DET: per-slot scale exponents
  yl.ex = exl;
  yl.xe = 0;
  if (DET) { // exponent of max |x| in the window: two words, tiles alternate
    if (threadIdx.x == 0) cfs_ticket[2] = cfs_ticket[3] = 0;
    __syncthreads();
  }
  int det_parity = 0;
  const int tid = threadIdx.x, lane = tid & 63;
  // wave-uniform by construction: keep it in an SGPR so that slice bookkeeping is
  // scalar (s_load / s_cbranch) and never waits on the vector-memory counter
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = BLOCK / 64;
  // blocks b and b+8 share an XCD (round-robin dispatch): give every XCD a
  // contiguous run of groups so neighbouring tiles share halo lines in one L2.
  // Placement only affects speed, never correctness.
  const int nper = gridDim.x >> 3;
  // (g is a launch SLOT: which group of the plan runs in it is the host's choice,
  // SymPlan::launch_order -- gfirst / group_range are stored in slot order)
  const int g = (blockIdx.x & 7) * nper + (blockIdx.x >> 3);
  const int2 trange = d.group_range[g];
  const int t0 = trange.x, t1 = trange.y;
  // developer timeline (cfs_hip_sym_debug_timeline): 100 MHz wall clock stamps of
  // this workgroup's phases; dbg is NULL in every product launch
  if (dbg && tid == 0) dbg[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memrealtime();

  // x window of a tile, gathered into registers with every load in flight at
  // once (two round trips: halo columns, then x).  Under a saturated memory
  // system a round trip costs microseconds, so the window of the NEXT tile is
  // requested before the barrier that ends the current tile's slices and lands
  // while the y window is flushed.
  // U = LDS slots one thread fills / flushes.  Every thread issues all U gathers
  // (clamped, unconditional), so a launch whose windows are small uses the
  // instantiation with the smallest U that covers them: a tile with 1 100 slots
  // would otherwise issue 78 % of its start-up loads for nothing.
  static_assert(U <= cfs_plan::kSlotsPerThread, "U");
  V xr[U];
  short er[DET ? U : 1]; // DET: the slots' scale exponents, fetched with the slot table
  bool first_gather = true;
  auto gather_x = [&](const Tile &tn) {
    // every load is unconditional (clamped index): a load under a divergent
    // branch would make the compiler drain vmcnt before the other side of the
    // branch may write the same register.  Own rows go through the slot table
    // too: tiles are clusters of the matrix graph, not runs of consecutive rows.
    int idx[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int i = min(tid + k * BLOCK, tn.nslots - 1);
      idx[k] = d.slot_col[tn.slot_off + i]; // slot_col is padded by one entry
      if (DET) er[k] = a_slot_exp[tn.slot_off + i];
    }
    if (dbg && tid == 0 && first_gather) { // diagnostic: when did the slot table arrive?
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      dbg[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime();
    }
    if (MODE != 4) {
#pragma unroll
      for (int k = 0; k < U; ++k) xr[k] = x[idx[k]];
    } else {
#pragma unroll
      for (int k = 0; k < U; ++k) xr[k] = V(idx[k]);
    }
  };
  // the group's first tile comes from a per-group copy: its descriptor does not
  // wait for group_ptr (one dependent round trip less before the first x gather)
  const Tile tfirst = d.gfirst[g];
  if (dbg && tid == 0) dbg[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime() + (tfirst.nown & 0);
  if (t0 < t1) gather_x(tfirst);
  first_gather = false;
  if (dbg && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dbg[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();
  }

  for (int ti = t0; ti < t1; ++ti) {
    const Tile t = ti == t0 ? tfirst : d.tiles[ti];
    const int nown = t.nown, nslots = t.nslots;
    const unsigned ny = OFFB ? (unsigned)t.ny : (unsigned)nslots; // slots with a y window
    const int vrow0 = t.vrow_off, nvr = t.nvrows;
    const V *tv = d.vals + t.nnz_off;
    const uint16_t *ts = d.slots + t.sl_off;
    const uint4 *smeta = d.slice_meta + t.slice_base;
    const int nsl = t.nslices;
    // leader lanes of this wave's first two slices: requested before the window
    // fill waits for the x gather, so that they are there when the head packet's
    // slot address needs them (one byte per lane; padded, clamped)
    const uint8_t *llp = d.leadlane + (size_t)t.slice_base * 64 + lane;
    int L_c = llp[min(wave, max(nsl, 1) - 1) * 64];
    int L_n = llp[min(wave + NW, max(nsl, 1) - 1) * 64];

    // fill both LDS windows.  A thread owns the same slot indices here and in the
    // flush at the end of the previous tile, so no barrier is needed between.
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int i = tid + k * BLOCK;
      if (i < nslots) {
        xl[i] = xr[k];
        yl.zero(i);
        if (DET) exl[i] = er[k];
      }
    }
    if (DET) { // biased exponent field of the largest |x| this thread brought in
      int ex = 0;
#pragma unroll
      for (int k = 0; k < U; ++k)
        if (tid + k * BLOCK < nslots) {
          const double ax = fabs((double)xr[k]);
          ex = max(ex, (int)((unsigned long long)__double_as_longlong(ax) >> 52));
        }
      // ... and of the x values its FAR entries (HYB) gather from outside the window: the
      // scale covers them too (a second pass over their columns, L2 hits later)
      for (int fp = wave; fp < ((t.nfar + 255) >> 8); fp += NW) {
        const int4 cc = *reinterpret_cast<const int4 *>(d.fcols + (size_t)t.far_off + (size_t)fp * 256u + lane * 4);
        const int e0 = fp * 256 + lane * 4;
        const int cq[4] = {cc.x, cc.y, cc.z, cc.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (e0 + u < t.nfar) {
            const double ax = fabs((double)x[cq[u]]);
            ex = max(ex, (int)((unsigned long long)__double_as_longlong(ax) >> 52));
          }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ex = max(ex, __shfl_xor(ex, o));
      if (lane == 0) atomicMax(&cfs_ticket[2 + det_parity], ex);
    }
    // the matrix stream does not depend on x: this wave's first slice header and
    // head packet (and its first COO packet) are requested BEFORE the barrier --
    // but after the window fill, so that the fill (and with it the barrier) waits
    // for the x gather only and not for these HBM loads.  Slices are handed out dynamically (they are sorted by cost, so
    // this is longest-first scheduling over the waves): a wave starts with
    // slices `wave` and `wave + NW` and draws every later one from an LDS
    // ticket counter one slice ahead of its use.
    int s = wave, s_n = wave + NW;
    uint4 meta_c = make_uint4(0u, 0u, 0u, 0u), meta_n = meta_c;
    uint32_t info_c = 0u;
    V dg_c = V(0);
    Pkt<V> N;
    N.v[0] = N.v[1] = N.v[2] = N.v[3] = V(0);
    N.c = make_ushort4(0, 0, 0, 0);
    int lr_c = 0; // rank of this lane's leader in the current slice
    if (s < nsl) {
      meta_c = smeta[s];
      if (s_n < nsl) meta_n = smeta[s_n];
      const int p0 = s * 64 + lane, q0 = min(p0, nvr - 1);
      const uint32_t i0 = stream_load<NT>(d.rowinfo + vrow0 + q0); // unconditional, clamped
      const V d0 = stream_load<NT>(d.diag + vrow0 + q0);
      info_c = p0 < nvr ? i0 : 0u;
      dg_c = p0 < nvr ? d0 : V(0);
      const unsigned long long lead0 = ((unsigned long long)meta_c.w << 32) | meta_c.z;
      lr_c = leader_rank(lead0, L_c & 63) | (L_c & 0xc0); // bits 6 / 7: sibling chain (entry_combined)
      fetch_packet<NT>(N, tv, ts, meta_c.x, meta_c.y & 0x1ffffffu, (int)(meta_c.y >> 25), lead0,
                       lane, lr_c & 63);
    }
    const int ncp = (t.ncoo + 255) >> 8; // COO packets of this tile
    Pkt<V> C;
    ushort4 Cr = make_ushort4(0, 0, 0, 0);
    C.v[0] = C.v[1] = C.v[2] = C.v[3] = V(0);
    C.c = Cr;
    if (wave < ncp) {
      fetch_packet<NT>(C, d.cvals + t.coo_off, d.ccols + t.coo_off, (uint32_t)wave * 256u,
                       (uint32_t)wave * 256u, 64, ~0ull, lane, lane);
      {
        const cfs_us4 rr = stream_load<NT>(reinterpret_cast<const cfs_us4 *>(d.crows + t.coo_off + wave * 256 + lane * 4));
        Cr = make_ushort4(rr.x, rr.y, rr.z, rr.w);
      }
    }
    if (tid == 0) cfs_ticket[0] = 2 * NW;
    __syncthreads();
    double det_unscale = 1.0;
    if (DET) {
      // |x| < 2^(E - 1022) in this window; the slots bring their own row bound (YWin)
      const int xe = __builtin_amdgcn_readfirstlane(cfs_ticket[2 + det_parity]);
      yl.xe = xe - 1022;
      // a NaN / Inf in the window of x (exponent field 2047) has no fixed-point image: every
      // row of the tile reads NaN, as the floating-point path would propagate it -- never
      // plausible finite garbage
      if (xe >= 2047) det_unscale = __longlong_as_double(0x7ff8000000000000ll);
      det_parity ^= 1;
      if (tid == 0) cfs_ticket[2 + det_parity] = 0; // the next tile's word (see the header comment)
    }
    if (dbg && tid == 0 && ti == t0) dbg[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memrealtime();

    while (MODE != 3 && s < nsl) {
      const uint32_t info = info_c;
      const V dg = dg_c;
      uint32_t off = meta_c.x, soff = meta_c.y & 0x1ffffffu;
      int cnt = (int)(meta_c.y >> 25);
      const unsigned long long leaders = ((unsigned long long)meta_c.w << 32) | meta_c.z;
      Pkt<V> A = N;
      const int lr = lr_c & 63; // leader rank of this lane in the current slice
      // COMB: one atomic per run of sibling lanes (entry_combined); a schedule with few such
      // runs -- one unknown per mesh node -- launches the plain instantiation
      const bool give = MODE == 0 && COMB && (lr_c & 64), take = MODE == 0 && COMB && (lr_c & 128);
      // ticket for the slice after next (its metadata is a scalar load that
      // lands long before it is needed)
      int s_nn = 0;
      if (lane == 0) s_nn = atomicAdd(&cfs_ticket[0], 1);
      s_nn = __builtin_amdgcn_readfirstlane(s_nn);
      const bool have_next = s_n < nsl;
      if (have_next) { // next slice: header + head packet, a whole slice ahead
        const int pn = s_n * 64 + lane, qn = min(pn, nvr - 1);
        const uint32_t in_ = stream_load<NT>(d.rowinfo + vrow0 + qn); // unconditional, clamped
        const V dn = stream_load<NT>(d.diag + vrow0 + qn);
        info_c = pn < nvr ? in_ : 0u;
        dg_c = pn < nvr ? dn : V(0);
        const unsigned long long lead_n = ((unsigned long long)meta_n.w << 32) | meta_n.z;
        lr_c = leader_rank(lead_n, L_n & 63) | (L_n & 0xc0); // L_n was requested a slice ago
        fetch_packet<NT>(N, tv, ts, meta_n.x, meta_n.y & 0x1ffffffu, (int)(meta_n.y >> 25), lead_n,
                         lane, lr_c & 63);
      }
      meta_c = meta_n;
      if (s_nn < nsl) meta_n = smeta[s_nn];
      L_n = llp[min(s_nn, nsl - 1) * 64]; // leader lanes of the slice after next
      const int s_cur = s;
      s = s_n;
      s_n = s_nn;

      const int r = info & 0xffffu;
      const int a = (int)(info >> 16); // packets of this lane's row
      const V xi = xl[r];
      V acc = V(0);
      const int amax = __builtin_amdgcn_readfirstlane(a); // rows are sorted: lane 0 is longest

      int g = 0;
      Pkt<V> B;
      auto consume = [&](const Pkt<V> &pk, bool act) {
        if (MODE == 0 && COMB) consume_combined<V, OFFB, DET>(pk, xl, yl, xi, acc, ny, act, give, take);
        else if (act) consume_packet<V, MODE, OFFB, DET>(pk, xl, yl, xi, acc, ny);
      };
      while (g + 2 < amax) { // steady state: two packets per trip, no copies
        const int cnt1 = __popcll(__ballot(a > g + 1));
        const uint32_t off1 = off + 4u * (uint32_t)cnt, soff1 = soff + slot_block(leaders, cnt);
        fetch_packet<NT>(B, tv, ts, off1, soff1, cnt1, leaders, lane, lr);
        consume(A, a > g);
        const int cnt2 = __popcll(__ballot(a > g + 2));
        const uint32_t off2 = off1 + 4u * (uint32_t)cnt1, soff2 = soff1 + slot_block(leaders, cnt1);
        fetch_packet<NT>(A, tv, ts, off2, soff2, cnt2, leaders, lane, lr);
        consume(B, a > g + 1);
        g += 2;
        off = off2;
        soff = soff2;
        cnt = cnt2;
      }
      if (amax - g == 2) {
        const int cnt1 = __popcll(__ballot(a > g + 1));
        fetch_packet<NT>(B, tv, ts, off + 4u * (uint32_t)cnt, soff + slot_block(leaders, cnt), cnt1,
                         leaders, lane, lr);
        consume(A, a > g);
        consume(B, a > g + 1);
      } else if (amax - g == 1) {
        consume(A, a > g);
      }
      if (s_cur * 64 + lane < nvr) yl.add(r, (double)fma(dg, xi, acc));
    }
    // COO leftovers: packet p = wave, wave + NW, ...; the first one was requested
    // at the top of the tile
    for (int cp = wave; MODE != 3 && cp < ncp; cp += NW) {
      const Pkt<V> Q = C;
      const ushort4 Qr = Cr;
      if (cp + NW < ncp) {
        fetch_packet<NT>(C, d.cvals + t.coo_off, d.ccols + t.coo_off, (uint32_t)(cp + NW) * 256u,
                         (uint32_t)(cp + NW) * 256u, 64, ~0ull, lane, lane);
        const cfs_us4 rr = stream_load<NT>(reinterpret_cast<const cfs_us4 *>(d.crows + t.coo_off + (cp + NW) * 256 + lane * 4));
        Cr = make_ushort4(rr.x, rr.y, rr.z, rr.w);
      }
      const int e0 = cp * 256 + lane * 4;
      if (e0 + 0 < t.ncoo) coo_update<V, MODE, OFFB, DET>(xl, yl, Q.v[0], Qr.x, Q.c.x, ny);
      if (e0 + 1 < t.ncoo) coo_update<V, MODE, OFFB, DET>(xl, yl, Q.v[1], Qr.y, Q.c.y, ny);
      if (e0 + 2 < t.ncoo) coo_update<V, MODE, OFFB, DET>(xl, yl, Q.v[2], Qr.z, Q.c.z, ny);
      if (e0 + 3 < t.ncoo) coo_update<V, MODE, OFFB, DET>(xl, yl, Q.v[3], Qr.w, Q.c.w, ny);
    }
    // FAR entries (HYB): a = A[row(r)][col] with col outside this tile and used only
    // once by it -- or the mirror image of such an entry of another tile.  One-sided:
    // y_l[r] += a * x[col], x gathered from global memory (L2): no slot, no strip
    for (int fp = wave; MODE != 3 && fp < ((t.nfar + 255) >> 8); fp += NW) {
      const size_t base = (size_t)t.far_off + (size_t)fp * 256u;
      V fv[4];
      if (sizeof(V) == 8) {
        const cfs_d2 lo = *reinterpret_cast<const cfs_d2 *>(d.fvals + base + lane * 2);
        const cfs_d2 hi = *reinterpret_cast<const cfs_d2 *>(d.fvals + base + 128 + lane * 2);
        fv[0] = lo.x; fv[1] = lo.y; fv[2] = hi.x; fv[3] = hi.y;
      } else {
        const cfs_f4 q4 = *reinterpret_cast<const cfs_f4 *>(d.fvals + base + lane * 4);
        fv[0] = q4.x; fv[1] = q4.y; fv[2] = q4.z; fv[3] = q4.w;
      }
      const cfs_us4 rr = *reinterpret_cast<const cfs_us4 *>(d.frows + base + lane * 4);
      const int4 cc = *reinterpret_cast<const int4 *>(d.fcols + base + lane * 4);
      const V x0 = x[cc.x], x1 = x[cc.y], x2 = x[cc.z], x3 = x[cc.w]; // padding: column 0
      const int e0 = fp * 256 + lane * 4;
      // (deterministic build: the tile's scale covers these x values too, see the window fill)
      if (e0 + 0 < t.nfar) yl.add(rr.x, (double)fv[0] * (double)x0);
      if (e0 + 1 < t.nfar) yl.add(rr.y, (double)fv[1] * (double)x1);
      if (e0 + 2 < t.nfar) yl.add(rr.z, (double)fv[2] * (double)x2);
      if (e0 + 3 < t.nfar) yl.add(rr.w, (double)fv[3] * (double)x3);
    }
    if (dbg && lane == 0 && wave == 0 && ti + 1 == t1) dbg[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();
    if (ti + 1 < t1) gather_x(d.tiles[ti + 1]); // lands behind the barrier + flush
    // y positions of the own rows (original numbering, block-local): requested
    // before the barrier, used by the flush after it
    int yidx[U];
#pragma unroll
    for (int k = 0; k < U; ++k)
      yidx[k] = d.slot_col[t.slot_off + min(tid + k * BLOCK, nown - 1)] - d.row_begin;
    __syncthreads();
    if (dbg && tid == 0 && ti + 1 == t1) dbg[blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memrealtime();
    // flush the y window: own rows -> y, halo sums -> this tile's strip.  Plain
    // stores; y is fully overwritten.
    if (MODE != 4) {
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int i = tid + k * BLOCK;
        if (i < nown) y[yidx[k]] = (V)yl.get(i, det_unscale);
        else if (i < (int)ny) d.strip[t.halo_off + (i - nown)] = (V)yl.get(i, det_unscale);
      }
    }
  }
  if (dbg && tid == 0) dbg[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
}

// halo fold: y[dst] += sum of the strip entries aimed at dst, in a fixed order.
// One 16-byte record per destination {dst, e0, e1, e2}: up to three strip entries
// are inlined (e1 / e2 = -1 when absent), so the common destination costs two
// dependent round trips: the record, then y[dst] and its entries all in flight
// together.  A longer list keeps e0, e1 in the record and e2 = -(offset + 2) of
// its remainder in `fidx`: [count, entry 2, entry 3, ...].  A lane sums the next
// 16 entries itself, four loads in flight at a time; what is left of a very long
// list (a hub row collects contributions from many tiles) is summed by the whole
// wave, strided, with a fixed shuffle tree -- one slow lane would otherwise
// decide the duration of the launch.  The order of the additions is fixed.
template <typename V>
__global__ void __launch_bounds__(256)
    cfs_fold_kernel(V *__restrict__ y, const V *__restrict__ src,
                    const int4 *__restrict__ frec, const int32_t *__restrict__ fidx, int m) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int r = 0, b = 0, e = 0;
  V s = V(0);
  if (i < m) {
    const int4 rec = frec[i];
    r = rec.x;
    const V y0 = y[r];
    const V s0 = src[rec.y];
    const V s1 = src[max(rec.z, 0)]; // unconditional, clamped: all in flight together
    const V s2 = src[max(rec.w, 0)];
    s = y0 + s0;
    if (rec.z >= 0) s += s1;
    if (rec.w >= 0) s += s2;
    if (rec.w < -1) {
      b = -(rec.w + 2);
      e = b + 1 + fidx[b];
      b += 1;
    }
  }
  const int own_end = min(e, b + 16);
  for (int q = b; q < own_end; q += 4) {
    int j[4];
    V v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) j[u] = fidx[min(q + u, own_end - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = src[j[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (q + u < own_end) s += v[u];
  }
  unsigned long long need = __ballot(e > own_end);
  while (need) {
    const int L = __ffsll((long long)need) - 1;
    need &= need - 1;
    const int lb = __shfl(own_end, L), le = __shfl(e, L);
    V p = V(0);
    for (int q = lb + lane; q < le; q += 64) p += src[fidx[q]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) p += __shfl_down(p, o);
    const V tot = __shfl(p, 0);
    if (lane == L) s += tot;
  }
  if (i < m) y[r] = s;
}

// new values into an existing schedule (cfs_hip_sym_update_values_*): every entry of a
// value array of the device format takes the caller's value at its recorded position
// (GPU-side packing of the values: the sparsity pattern, and with it the whole schedule
// -- tiles, slots, leaders, fold index -- stays)
template <typename V>
__global__ void __launch_bounds__(256)
    cfs_value_scatter_kernel(V *__restrict__ dst, const int32_t *__restrict__ map,
                             const V *__restrict__ src, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int32_t m = map[i];
    dst[i] = m >= 0 ? src[m] : V(0);
  }
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

// the dense form of the exchange (north-star: reduce-scatter of the off-block contributions):
// a shard's packed contributions go to their slot of a zeroed vector of nranks equal blocks ...
template <typename V>
__global__ void __launch_bounds__(256)
    cfs_scatter_pos_kernel(V *__restrict__ dense, const int32_t *__restrict__ pos, const V *__restrict__ packed, int m) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) dense[pos[i]] = packed[i];
}
// ... and the block a rank receives is added to its rows
template <typename V>
__global__ void __launch_bounds__(256) cfs_add_rows_kernel(V *__restrict__ y, const V *__restrict__ add, int m) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) y[i] += add[i];
}

// ---------------------------------------------------------------------------
// host objects
// ---------------------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the kernel FUNCTION (one
// template instantiation, per device), not to a handle: two handles with different
// windows share instantiations.  Every instantiation is raised ONCE per device to
// the ceiling of a CU (160 KiB minus the static ticket counter) and never lowered.
static int raise_lds_limit(const void *kernel, int device) {
  static std::mutex mu;
  static std::vector<std::pair<const void *, int>> done;
  std::lock_guard<std::mutex> lk(mu);
  for (auto &d : done)
    if (d.first == kernel && d.second == device) return 0;
  HIPCHK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                             160 * 1024 - 64));
  done.push_back({kernel, device});
  return 0;
}

// fold records {dst, e0, e1, e2 | -(offset + 2)} and the remainder lists
// [count, entries 2..] of destinations with more than three contributions
// (see cfs_fold_kernel)
static void make_fold_records(const std::vector<int32_t> &dst, const std::vector<int32_t> &ptr,
                              const std::vector<int32_t> &idx, std::vector<int4> &rec,
                              std::vector<int32_t> &rest) {
  rec.assign(dst.size() + 1, make_int4(0, 0, -1, -1));
  rest.clear();
  for (size_t i = 0; i < dst.size(); i++) {
    const int b = ptr[i], len = ptr[i + 1] - ptr[i];
    int4 r = make_int4(dst[i], idx[b], len > 1 ? idx[b + 1] : -1, len == 3 ? idx[b + 2] : -1);
    if (len > 3) {
      r.w = -((int)rest.size() + 2);
      rest.push_back(len - 2);
      rest.insert(rest.end(), idx.begin() + b + 2, idx.begin() + b + len);
    }
    rec[i] = r;
  }
}

// staging of a handle for callers that pass HOST pointers (the reference's API
// hands raw host pointers every call, include/kernel/sparse_kernel.hpp:22-23):
// device mirrors of x / y, and page-locked blocks from the pinned pool
// (CFS_HIP_MEM_PINNED, allocated once per handle) that pageable vectors are
// copied through with all host threads.  A vector that already lives in
// page-locked memory (internal_alloc(.., Platform::cpu) of this build hands such
// blocks out) is DMA-ed in place.
struct HostStage {
  DevBuf xdev, ydev;
  PinBuf xpin, ypin;
};

// y <- launch(x) for x / y that may be host or device pointers, on `device`'s
// library stream.  Returns after the result is complete when host memory is
// involved; with both vectors resident the work is only enqueued.
template <class Launch>
static int run_staged(int device, HostStage &S, size_t xbytes, size_t ybytes, void *y,
                      const void *x, Launch launch) {
  cfs_rt::DevCtx *ctx;
  int rc = cfs_rt::device_ctx(device, &ctx);
  if (rc) return rc;
  hipStream_t st = ctx->stream;
  DeviceGuard g(device);
  const cfs_rt::PtrInfo xi = cfs_rt::classify(x), yi = cfs_rt::classify(y);
  if ((xi.device && xi.dev != device) || (yi.device && yi.dev != device))
    return set_err(CFS_HIP_ERR_ARG, "x / y live on device " +
                                        std::to_string(xi.device && xi.dev != device ? xi.dev : yi.dev) +
                                        ", the matrix on device " + std::to_string(device));
  const void *xdev = x;
  void *ydev = y;
  if (!xi.device) {
    if (S.xdev.bytes < xbytes && (rc = S.xdev.alloc(xbytes))) return rc;
    const void *src = x;
    if (!xi.pinned) { // pageable: through the handle's page-locked block
      if ((rc = S.xpin.reserve(xbytes))) return rc;
      cfs_rt::parallel_copy(S.xpin.p, x, xbytes);
      src = S.xpin.p;
    }
    HIPCHK(hipMemcpyAsync(S.xdev.p, src, xbytes, hipMemcpyHostToDevice, st));
    xdev = S.xdev.p;
  }
  if (!yi.device) {
    if (S.ydev.bytes < ybytes && (rc = S.ydev.alloc(ybytes))) return rc;
    ydev = S.ydev.p;
  }
  if ((rc = launch(ydev, xdev, st))) return rc;
  if (!yi.device) {
    if (yi.pinned) {
      HIPCHK(hipMemcpyAsync(y, ydev, ybytes, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
    } else {
      if ((rc = S.ypin.reserve(ybytes))) return rc;
      HIPCHK(hipMemcpyAsync(S.ypin.p, ydev, ybytes, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      cfs_rt::parallel_copy(y, S.ypin.p, ybytes);
    }
  } else if (!xi.device) {
    HIPCHK(hipStreamSynchronize(st)); // the staged x may be overwritten by the next call
  }
  // both resident: enqueued on the library stream (a BLOCKING stream: a caller's
  // own hipMemcpy / hipDeviceSynchronize on the null stream orders behind it, and
  // so do cfs_hip_memcpy / cfs_hip_synchronize)
  return 0;
}

struct cfs_hip_sym_s {
  int value_bytes = 8;
  virtual ~cfs_hip_sym_s() {}
  virtual int spmv_local(void *y, const void *x, void *send, hipStream_t st, int phases = 7) = 0;
  virtual int recv_fold(void *y, const void *recv, hipStream_t st) = 0;
  virtual int set_recv(int nrecv, const int *rows) = 0;
  virtual void stats(cfs_hip_sym_stats *o) = 0;
  virtual const std::vector<int32_t> &send_counts() = 0;
  virtual const std::vector<int32_t> &send_rows() = 0;
  virtual int n() = 0;
  virtual int rows() = 0;
  virtual int timeline(void *y, const void *x, unsigned long long *host, int cap, int *ngroups) = 0;
  virtual int group_features(long long *out, int cap, int *ngroups) = 0;
  // values_dev: the caller's full CSR value array (same pattern as at create), on this device
  virtual int update_values(const void *values_dev, long long nnz, hipStream_t st) = 0;
  int device = 0; // the device this handle's arrays live on (current device at create)
  std::string plan_note; // why the device builder handed the schedule to the host builder ("" = it did not)
  HostStage stage; // host-pointer callers
  // the last (y, x) pair whose placement was validated (async entry points)
  const void *ok_x = nullptr, *ok_y = nullptr;
};

template <typename V> struct SymMatrix : cfs_hip_sym_s {
  SymPlan<V> P; // big arrays released after upload
  DevBuf tiles, gfirst, group_ptr, slot_col, rowinfo, diag, slice_meta, leadlane, vals, slots, strip;
  DevBuf cvals, crows, ccols, fvals, frows, fcols;
  DevBuf val_map, cval_map, fval_map, diag_map; // CFS_HIP_FLAG_KEEP_VALUE_MAP
  bool has_value_map = false;
  int64_t nnz_caller = 0; // entries of the caller's CSR the maps point into
  DevBuf fold_rec, fold_idx, send_ptr, send_idx;
  DevBuf slot_exp; // deterministic build: scale exponent of every slot
  const short *dev_slot_exp = nullptr;
  DevBuf rfold_rec, rfold_idx;
  SymDev<V> dev{};
  int nfold = 0, nsend = 0, nrfold = 0;
  int ablate_mode = 0; // cfs_hip_options.flags & 7 (timing-only ablations)
  bool nt_stream = true; // matrix stream larger than the Infinity Cache: non-temporal loads
  bool offblock = false; // some tile has one-sided (off-block) slots: mirrored shard
  bool combine_ok = false; // enough sibling chains (>= 10 % of the lane-packets) for the combining kernel
  bool combine = false;    // ... and it is the one this handle launches
  bool combine_forced = false;
  int64_t mirror_entries = 0;
  unsigned long long *dbg_buf = nullptr; // set only by cfs_hip_sym_debug_timeline
  size_t lds_bytes = 0;
  int64_t halo_slots = 0, stream_len = 0, slot_len = 0, nslices = 0, coo_len = 0, far_len = 0, far_entries = 0;

  bool device_built = false; // the schedule was built by cfs_devplan.hpp (arrays never on the host)

  int upload() {
    int rc;
#define UP(buf, vec)                                                          \
  if ((rc = buf.upload(vec.data(), vec.size() * sizeof(vec[0])))) return rc;
    UP(tiles, P.tiles)
    UP(slot_col, P.slot_col)
    UP(rowinfo, P.rowinfo)
    UP(diag, P.diag)
    UP(slice_meta, P.slice_meta)
    UP(leadlane, P.leadlane)
    UP(cvals, P.cvals)
    UP(crows, P.crows)
    UP(ccols, P.ccols)
    UP(vals, P.vals)
    UP(slots, P.slots)
    UP(fvals, P.fvals)
    UP(frows, P.frows)
    UP(fcols, P.fcols)
    has_value_map = !P.val_map.empty();
    if (has_value_map) {
      UP(val_map, P.val_map)
      UP(cval_map, P.cval_map)
      UP(fval_map, P.fval_map)
      UP(diag_map, P.diag_map)
      cfs_plan::release_async(P.val_map);
      std::vector<int32_t>().swap(P.cval_map);
      std::vector<int32_t>().swap(P.fval_map);
      std::vector<int32_t>().swap(P.diag_map);
    }
    {
      std::vector<int4> rec;
      std::vector<int32_t> rest;
      make_fold_records(P.fold_dst, P.fold_ptr, P.fold_idx, rec, rest);
      if ((rc = fold_rec.upload(rec.data(), rec.size() * sizeof(int4)))) return rc;
      if ((rc = fold_idx.upload(rest.data(), rest.size() * 4))) return rc;
    }
    UP(send_ptr, P.send_ptr)
    UP(send_idx, P.send_idx)
    if (P.deterministic) {
      UP(slot_exp, P.slot_exp)
      dev_slot_exp = (const short *)slot_exp.p;
    }
#undef UP
    if ((rc = finish_setup((int64_t)P.slice_meta.size()))) return rc;
    // release the big host arrays; keep the small metadata
    cfs_plan::release_async(P.vals);
    cfs_plan::release(P.slots);
    std::vector<uint8_t>().swap(P.leadlane);
    std::vector<V>().swap(P.cvals);
    std::vector<uint16_t>().swap(P.crows);
    std::vector<uint16_t>().swap(P.ccols);
    std::vector<V>().swap(P.fvals);
    std::vector<uint16_t>().swap(P.frows);
    std::vector<int32_t>().swap(P.fcols);
    std::vector<V>().swap(P.diag);
    std::vector<uint32_t>().swap(P.rowinfo);
    std::vector<int32_t>().swap(P.fold_idx);
    std::vector<int32_t>().swap(P.send_idx);
    return 0;
  }
  // the schedule's arrays were built in place by the device builder: only the launch
  // tables and the derived settings are left to do
  int adopt_device_schedule() {
    device_built = true;
    int64_t ns = 0;
    for (const Tile &t : P.tiles) ns += t.nslices;
    return finish_setup(ns);
  }

  // what both builders share once the arrays are on the device: launch-slot tables, strips,
  // kernel choice, LDS window, staging room, cache policy
  int finish_setup(int64_t nslices_in) {
    int rc;
    { // per launch slot: first tile + tile range of the group that runs there
      const int G = (int)P.group_first.size();
      std::vector<Tile> gf(G);
      std::vector<int2> gr(G);
      for (int sl = 0; sl < G; sl++) {
        const int g = (int)P.launch_order.size() == G ? P.launch_order[sl] : sl;
        gf[sl] = P.group_first[g];
        gr[sl] = make_int2(P.group_ptr[g], P.group_ptr[g + 1]);
      }
      if ((rc = gfirst.upload(gf.data(), gf.size() * sizeof(Tile)))) return rc;
      if ((rc = group_ptr.upload(gr.data(), gr.size() * sizeof(int2)))) return rc;
    }
    if ((rc = strip.alloc((size_t)P.nhalo * sizeof(V)))) return rc;
    halo_slots = P.nhalo;
    offblock = P.onesided_slots > 0;
    // combining kernel: enough sibling chains, and a launch long enough for the longer
    // dependency chain per entry to hide (measured: ldoor stand-in, 283 MB per launch,
    // +3 %; 1/8 shards, 89-131 MB, -8 %); tune() times both where it may (sym_create)
    combine_ok = P.chained_packets * 10 >= P.lane_packets && P.chained_packets > 0;
    combine = combine_ok && P.stream_len * (int64_t)sizeof(V) + P.slot_len * 2 >= (int64_t)200 * 1000 * 1000;
    combine_forced = false;
    if (const char *e = getenv("CFS_HIP_COMBINE")) // developer knob: 2 = the combining kernel whatever the size
      if (atoi(e) == 2) combine = combine_ok, combine_forced = true;
    mirror_entries = P.mirror_entries;
    stream_len = P.stream_len;
    slot_len = P.slot_len;
    nslices = nslices_in;
    coo_len = P.coo_len;
    far_len = P.far_len;
    far_entries = P.far_entries;
    nfold = (int)P.fold_dst.size();
    nsend = (int)P.send_row.size();
    dev.tiles = (const Tile *)tiles.p;
    dev.gfirst = (const Tile *)gfirst.p;
    dev.group_range = (const int2 *)group_ptr.p;
    dev.slot_col = (const int32_t *)slot_col.p;
    dev.rowinfo = (const uint32_t *)rowinfo.p;
    dev.diag = (const V *)diag.p;
    dev.slice_meta = (const uint4 *)slice_meta.p;
    dev.leadlane = (const uint8_t *)leadlane.p;
    dev.cvals = (const V *)cvals.p;
    dev.crows = (const uint16_t *)crows.p;
    dev.ccols = (const uint16_t *)ccols.p;
    dev.vals = (const V *)vals.p;
    dev.slots = (const uint16_t *)slots.p;
    dev.fvals = (const V *)fvals.p;
    dev.frows = (const uint16_t *)frows.p;
    dev.fcols = (const int32_t *)fcols.p;
    dev.strip = (V *)strip.p;
    dev.row_begin = P.row_begin;
    dev.lds_slots = P.lds_slots;
    lds_bytes = (size_t)P.lds_slots * (size_t)cfs_plan::slot_lds_bytes<V>(P.deterministic);
    // the stream is cacheable across SpMVs only if it fits the 256 MiB Infinity Cache
    nt_stream = (stream_len * (int64_t)sizeof(V) + slot_len * 2) > (int64_t)240 * 1024 * 1024;
    if (getenv("CFS_PLAN_VERBOSE"))
      fprintf(stderr, "[cfs_hip] handle: %s-built, %d tiles, window %d slots, %d threads x %d per CU, sibling chains %lld of %lld "
              "lane-packets -> %s kernel, %s stream loads\n", device_built ? "device" : "host", (int)P.tiles.size(),
              P.lds_slots, P.block_threads, P.wg_per_cu, (long long)P.chained_packets, (long long)P.lane_packets,
              combine ? "combining" : "plain", nt_stream ? "non-temporal" : "cacheable");
    return 0;
  }

  // the instantiation of the tile kernel this handle launches
  template <int BLOCK> static const void *pick_kernel(int mode, bool nt, bool offb, int u, bool det, bool comb) {
#define CFS_K(M, N, O, UU) ((const void *)cfs_sym_tile_kernel<V, BLOCK, M, N, O, UU>)
#define CFS_KP(N, O, UU) ((const void *)cfs_sym_tile_kernel<V, BLOCK, 0, N, O, UU, false, false>)
#define CFS_KDET(N, O, UU) ((const void *)cfs_sym_tile_kernel<V, (BLOCK < 512 ? 512 : BLOCK), 0, N, O, UU, true>)
    constexpr int UM = cfs_plan::kSlotsPerThread;
    switch (mode) {
    case 1: return CFS_K(1, true, false, UM);
    case 2: return CFS_K(2, true, false, UM);
    case 3: return CFS_K(3, true, false, UM);
    case 4: return CFS_K(4, true, false, UM);
    default: break;
    }
    if (det) { // CFS_HIP_FLAG_DETERMINISTIC: 512 or 1 024 threads (to_opts)
      static const void *const dtab[2][2][3] = {
          {{CFS_KDET(false, false, 3), CFS_KDET(false, false, 6), CFS_KDET(false, false, UM)},
           {CFS_KDET(false, true, 3), CFS_KDET(false, true, 6), CFS_KDET(false, true, UM)}},
          {{CFS_KDET(true, false, 3), CFS_KDET(true, false, 6), CFS_KDET(true, false, UM)},
           {CFS_KDET(true, true, 3), CFS_KDET(true, true, 6), CFS_KDET(true, true, UM)}}};
      return dtab[nt ? 1 : 0][offb ? 1 : 0][u <= 3 ? 0 : (u <= 6 ? 1 : 2)];
    }
    static const void *const tab[2][2][3] = {
        {{CFS_K(0, false, false, 3), CFS_K(0, false, false, 6), CFS_K(0, false, false, UM)},
         {CFS_K(0, false, true, 3), CFS_K(0, false, true, 6), CFS_K(0, false, true, UM)}},
        {{CFS_K(0, true, false, 3), CFS_K(0, true, false, 6), CFS_K(0, true, false, UM)},
         {CFS_K(0, true, true, 3), CFS_K(0, true, true, 6), CFS_K(0, true, true, UM)}}};
    static const void *const ptab[2][2][3] = { // plain: no sibling-combined atomics
        {{CFS_KP(false, false, 3), CFS_KP(false, false, 6), CFS_KP(false, false, UM)},
         {CFS_KP(false, true, 3), CFS_KP(false, true, 6), CFS_KP(false, true, UM)}},
        {{CFS_KP(true, false, 3), CFS_KP(true, false, 6), CFS_KP(true, false, UM)},
         {CFS_KP(true, true, 3), CFS_KP(true, true, 6), CFS_KP(true, true, UM)}}};
#undef CFS_K
#undef CFS_KP
#undef CFS_KDET
    return (comb ? tab : ptab)[nt ? 1 : 0][offb ? 1 : 0][u <= 3 ? 0 : (u <= 6 ? 1 : 2)];
  }
  const void *tile_kernel() {
    const int u = (P.lds_slots + P.block_threads - 1) / P.block_threads;
    switch (P.block_threads) {
    case 256: return pick_kernel<256>(ablate_mode, nt_stream, offblock, u, false, combine);
    case 512: return pick_kernel<512>(P.deterministic ? 0 : ablate_mode, nt_stream, offblock, u,
                                      P.deterministic, combine);
    default: return pick_kernel<1024>(P.deterministic ? 0 : ablate_mode, nt_stream, offblock, u, P.deterministic,
                                      combine);
    }
  }
  int launch_tiles(V *y, const V *x, hipStream_t st) {
    const void *k = tile_kernel();
    int rc = raise_lds_limit(k, device);
    if (rc) return rc;
    void *args[] = {(void *)&dev.tiles, (void *)&dev.gfirst, (void *)&dev.group_range,
                    (void *)&dev.slot_col, (void *)&dev.rowinfo, (void *)&dev.diag,
                    (void *)&dev.slice_meta, (void *)&dev.leadlane, (void *)&dev.vals,
                    (void *)&dev.slots,
                    (void *)&dev.cvals, (void *)&dev.crows, (void *)&dev.ccols,
                    (void *)&dev.fvals, (void *)&dev.frows, (void *)&dev.fcols,
                    (void *)&dev.strip, (void *)&dev.row_begin, (void *)&dev.lds_slots,
                    (void *)&x, (void *)&y, (void *)&dbg_buf, (void *)&dev_slot_exp};
    HIPCHK(hipLaunchKernel(k, dim3(P.ngroups), dim3(P.block_threads), args, lds_bytes, st));
    return 0;
  }

  int spmv_local(void *yv, const void *xv, void *sendv, hipStream_t st, int phases) override {
    V *y = (V *)yv;
    const V *x = (const V *)xv;
    if (P.tiles.empty()) return 0;
    if (phases & CFS_HIP_PHASE_TILES) {
      int rc = launch_tiles(y, x, st);
      if (rc) return rc;
    }
    // pack first: the exchange of a shard can then start while the local fold runs
    if ((phases & CFS_HIP_PHASE_PACK) && nsend > 0) {
      if (!sendv) return set_err(CFS_HIP_ERR_ARG, "shard has remote rows: send_buf required");
      hipLaunchKernelGGL((cfs_pack_kernel<V>), dim3((nsend + 255) / 256), dim3(256), 0,
                         st, (V *)sendv, (const V *)strip.p, (const int32_t *)send_ptr.p,
                         (const int32_t *)send_idx.p, nsend);
    }
    if ((phases & CFS_HIP_PHASE_FOLD) && nfold > 0)
      hipLaunchKernelGGL((cfs_fold_kernel<V>), dim3((nfold + 255) / 256), dim3(256), 0,
                         st, y, (const V *)strip.p, (const int4 *)fold_rec.p,
                         (const int32_t *)fold_idx.p, nfold);
    HIPCHK(hipGetLastError());
    return 0;
  }

  int set_recv(int nrecv, const int *rows) override {
    if (!cfs_plan::set_recv(P, nrecv, rows)) return set_err(CFS_HIP_ERR_ARG, P.error);
    int rc;
    rfold_rec = DevBuf();
    rfold_idx = DevBuf();
    {
      std::vector<int4> rec;
      std::vector<int32_t> rest;
      make_fold_records(P.rfold_row, P.rfold_ptr, P.rfold_idx, rec, rest);
      if ((rc = rfold_rec.upload(rec.data(), rec.size() * sizeof(int4)))) return rc;
      if ((rc = rfold_idx.upload(rest.data(), rest.size() * 4))) return rc;
    }
    nrfold = (int)P.rfold_row.size();
    return 0;
  }

  int recv_fold(void *yv, const void *recv, hipStream_t st) override {
    if (nrfold > 0)
      hipLaunchKernelGGL((cfs_fold_kernel<V>), dim3((nrfold + 255) / 256), dim3(256), 0,
                         st, (V *)yv, (const V *)recv, (const int4 *)rfold_rec.p,
                         (const int32_t *)rfold_idx.p, nrfold);
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
    o->mirror_entries = mirror_entries;
    o->bytes_algorithmic = P.nnz_low * (4 + s) + rows_ * (4 + 3 * s);
    o->far_entries = far_entries;
    o->ngroups = P.ngroups;
    o->bytes_streamed = stream_len * s + slot_len * 2 + coo_len * (s + 4) + far_len * (2 * s + 6) +
                        rows_ * (4 + 3 * s) +
                        halo_slots * (4 + 2 * s) + rows_ * 8 /* slot_col, x, strip st */
                        + halo_slots * (4 + s)              /* fold: idx + strip ld  */
                        + (int64_t)(nfold + nsend) * (8 + 2 * s) + nslices * 16 +
                        (int64_t)P.tiles.size() * (int64_t)sizeof(Tile);
    o->device_bytes = (int64_t)(tiles.bytes + group_ptr.bytes + slot_col.bytes +
                                rowinfo.bytes + diag.bytes + slice_meta.bytes + leadlane.bytes + vals.bytes + cvals.bytes + crows.bytes + ccols.bytes +
                                fvals.bytes + frows.bytes + fcols.bytes + val_map.bytes + cval_map.bytes +
                                fval_map.bytes + diag_map.bytes +
                                slots.bytes + strip.bytes + fold_rec.bytes +
                                fold_idx.bytes + send_ptr.bytes + send_idx.bytes);
  }
  const std::vector<int32_t> &send_counts() override { return P.send_counts; }
  const std::vector<int32_t> &send_rows() override { return P.send_row; }
  int timeline(void *y, const void *x, unsigned long long *host, int cap, int *ng) override {
    *ng = P.ngroups;
    if (cap < P.ngroups * 8) return set_err(CFS_HIP_ERR_ARG, "buffer too small: need 8 words per group");
    DevBuf b;
    int rc = b.alloc((size_t)P.ngroups * 8 * sizeof(unsigned long long));
    if (rc) return rc;
    HIPCHK(hipMemset(b.p, 0, b.bytes));
    for (int it = 0; it < 3; it++) { // warm, then the recorded launch
      dbg_buf = it == 2 ? (unsigned long long *)b.p : nullptr;
      rc = spmv_local(y, x, nullptr, (hipStream_t)0, CFS_HIP_PHASE_TILES);
      dbg_buf = nullptr;
      if (rc) return rc;
    }
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(host, b.p, b.bytes, hipMemcpyDeviceToHost));
    return 0;
  }
  // per persistent group, kCfsGroupFeatures words (cfs_hip_sym_debug_group_features)
  int group_features(long long *out, int cap, int *ng) override {
    *ng = P.ngroups;
    if (cap < P.ngroups * CFS_HIP_GROUP_FEATURES)
      return set_err(CFS_HIP_ERR_ARG, "buffer too small");
    const int T = (int)P.tiles.size();
    for (int sl = 0; sl < P.ngroups; sl++) { // in launch-slot order
      const int g = (int)P.launch_order.size() == P.ngroups ? P.launch_order[sl] : sl;
      long long *o = out + (size_t)sl * CFS_HIP_GROUP_FEATURES;
      for (int k = 0; k < CFS_HIP_GROUP_FEATURES; k++) o[k] = 0;
      for (int ti = P.group_ptr[g]; ti < P.group_ptr[g + 1]; ti++) {
        const Tile &t = P.tiles[ti];
        o[0] += 1;
        o[1] += t.nown;
        o[2] += t.nvrows;
        o[3] += t.nslices;
        o[4] += ti < (int)P.tile_rounds.size() ? P.tile_rounds[ti] : 0;
        o[5] += (ti + 1 < T ? P.tiles[ti + 1].nnz_off : stream_len) - t.nnz_off;
        o[6] += (ti + 1 < T ? P.tiles[ti + 1].sl_off : slot_len) - t.sl_off;
        o[7] += t.ncoo;
        o[8] += t.nslots - t.nown;
        o[9] += t.nslots;
      }
    }
    return 0;
  }
  int update_values(const void *values_dev, long long nnz, hipStream_t st) override {
    if (!has_value_map)
      return set_err(CFS_HIP_ERR_ARG, "handle was created without CFS_HIP_FLAG_KEEP_VALUE_MAP");
    if (P.deterministic)
      return set_err(CFS_HIP_ERR_UNSUPPORTED, "deterministic handle: its fixed-point scale depends on the values");
    if (nnz != nnz_caller) return set_err(CFS_HIP_ERR_ARG, "value count differs from the matrix the handle was built from");
    const V *src = (const V *)values_dev;
    auto go = [&](DevBuf &dst, DevBuf &map) {
      const long long cnt = (long long)(map.bytes / 4);
      if (cnt <= 0) return;
      const int grid = (int)std::min<long long>((cnt + 255) / 256, 256 * 16);
      hipLaunchKernelGGL((cfs_value_scatter_kernel<V>), dim3(grid), dim3(256), 0, st, (V *)dst.p,
                         (const int32_t *)map.p, src, cnt);
    };
    go(vals, val_map);
    go(cvals, cval_map);
    go(fvals, fval_map);
    go(diag, diag_map);
    HIPCHK(hipGetLastError());
    return 0;
  }
  int n() override { return P.n; }
  int rows() override { return P.row_end - P.row_begin; }
};

#include "cfs_devplan.hpp" // tune() on the GPU (needs SymMatrix and cfs_value_scatter_kernel)
#include "cfs_csr.hpp"     // the general CSR path: kernels, handle, create / launch
#include "cfs_solver.hpp"  // conjugate gradients on resident vectors (a solver-style caller)


// ---------------------------------------------------------------------------
// C ABI (every function below is declared extern "C" in cfs_hip.h)
// ---------------------------------------------------------------------------

int cfs_hip_abi_version(void) { return CFS_HIP_ABI_VERSION; }
const char *cfs_hip_last_error(void) { return cfs_rt::last_error().c_str(); }

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

// Make `device` the calling thread's current device and the home of the
// synchronous entry points.  Contexts (streams) of other devices stay alive:
// handles created on them keep working.
int cfs_hip_init(int device) { return cfs_rt::bind_home(device); }

int cfs_hip_current_device(int *device) {
  if (!device) return set_err(CFS_HIP_ERR_ARG, "device is NULL");
  int rc = ensure_init();
  if (rc) return rc;
  *device = cfs_rt::rt().home.load();
  return 0;
}

// 1 once a home device is bound (cfs_hip_init, or the first entry point that needed one);
// never initialises anything itself
int cfs_hip_runtime_bound(void) { return cfs_rt::rt().home.load() >= 0 ? 1 : 0; }

int cfs_hip_default_stream(void **stream) {
  int rc = ensure_init();
  if (rc) return rc;
  *stream = (void *)cfs_rt::home_stream();
  return 0;
}

int cfs_hip_synchronize(void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  if (stream) {
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
  }
  // NULL: the library streams of the synchronous entry points, on every device
  // a handle of this process lives on (a matrix sharded over N GPUs by the C++
  // surface runs on N of them)
  for (int d = 0; d < cfs_rt::kMaxDevices; d++) {
    hipStream_t st;
    {
      std::lock_guard<std::mutex> lk(cfs_rt::rt().mu);
      st = cfs_rt::rt().ctx[d].stream;
    }
    if (!st) continue;
    DeviceGuard g(d);
    HIPCHK(hipStreamSynchronize(st));
  }
  return 0;
}

int cfs_hip_alloc(size_t bytes, int kind, void **out) {
  if (!out) return set_err(CFS_HIP_ERR_ARG, "out is NULL");
  int rc = ensure_init();
  if (rc) return rc;
  if (bytes == 0) bytes = 64;
  if (kind == CFS_HIP_MEM_DEVICE) {
    DeviceGuard g(cfs_rt::rt().home.load());
    HIPCHK(hipMalloc(out, bytes));
  } else if (kind == CFS_HIP_MEM_PINNED) {
    return cfs_rt::pinned().alloc(bytes, out);
  } else {
    return set_err(CFS_HIP_ERR_ARG, "unknown memory kind");
  }
  return 0;
}

int cfs_hip_free(void *p, int kind) {
  if (!p) return 0;
  if (kind == CFS_HIP_MEM_DEVICE) {
    // (hipFree finds the owning device itself; the guard keeps the calling thread's
    // current device what it was on runtimes that switch to the owner)
    const cfs_rt::PtrInfo pi = cfs_rt::classify(p);
    DeviceGuard g(pi.device ? pi.dev : -1);
    HIPCHK(hipFree(p));
  }
  else if (kind == CFS_HIP_MEM_PINNED) return cfs_rt::pinned().release(p);
  else return set_err(CFS_HIP_ERR_ARG, "unknown memory kind");
  return 0;
}

int cfs_hip_pinned_owns(const void *p) { return p && cfs_rt::pinned().owns(p) ? 1 : 0; }

int cfs_hip_pinned_pool_stats(size_t *live_blocks, size_t *spare_blocks, size_t *spare_bytes) {
  size_t a = 0, b = 0, c = 0;
  cfs_rt::pinned().stats(&a, &b, &c);
  if (live_blocks) *live_blocks = a;
  if (spare_blocks) *spare_blocks = b;
  if (spare_bytes) *spare_bytes = c;
  return 0;
}

int cfs_hip_memcpy(void *dst, const void *src, size_t bytes, int dir) {
  int rc = ensure_init();
  if (rc) return rc;
  hipMemcpyKind k = dir == CFS_HIP_H2D   ? hipMemcpyHostToDevice
                    : dir == CFS_HIP_D2H ? hipMemcpyDeviceToHost
                                         : hipMemcpyDeviceToDevice;
  if ((rc = cfs_hip_synchronize(nullptr))) return rc; // pending SpMVs on resident vectors
  HIPCHK(hipMemcpy(dst, src, bytes, k));
  return 0;
}

int cfs_hip_memset(void *dst, int value, size_t bytes) {
  if (!dst) return set_err(CFS_HIP_ERR_ARG, "dst is NULL");
  int rc = ensure_init();
  if (rc) return rc;
  const cfs_rt::PtrInfo pi = cfs_rt::classify(dst);
  DeviceGuard g(pi.device ? pi.dev : -1);
  if ((rc = cfs_hip_synchronize(nullptr))) return rc; // pending SpMVs on resident vectors
  HIPCHK(hipMemset(dst, value, bytes));
  return 0;
}

// what a refusal of the schedule builder means to the caller.  UNSUPPORTED is
// reserved for matrices the tile schedule does not cover (a row denser than an LDS
// window, 16-/25-/31-bit offsets exhausted): src/csr.cpp then binds the general CSR
// kernel.  Bad arguments, the mirror refusals of a shard and builder bugs get codes
// of their own -- they must never turn into a quietly slower path.
static int plan_error_code(const std::string &e) {
  if (e.rfind("mirror:", 0) == 0) return CFS_HIP_ERR_MIRROR;
  if (e.rfind("internal:", 0) == 0) return CFS_HIP_ERR_INTERNAL;
  if (e.rfind("bad ", 0) == 0 || e.rfind("block_threads", 0) == 0) return CFS_HIP_ERR_ARG;
  return CFS_HIP_ERR_UNSUPPORTED;
}

static cfs_plan::Options to_opts(const cfs_hip_options *o) {
  cfs_plan::Options r;
  if (o) {
    r.max_slots = o->max_slots;
    r.max_tile_nnz = o->max_tile_nnz;
    r.block_threads = o->block_threads;
    r.flags = o->flags;
    r.reorder = !(o->flags & CFS_HIP_FLAG_NO_REORDER);
    if (o->flags & CFS_HIP_FLAG_FORCE_CLUSTER) r.force_order = 2;
    if (o->flags & CFS_HIP_FLAG_SHARD_EXCHANGE) r.mirror_offblock = false;
    if (o->flags & CFS_HIP_FLAG_HYB) r.hyb = true;
    if (o->flags & CFS_HIP_FLAG_DETERMINISTIC) r.deterministic = true;
    if (o->flags & CFS_HIP_FLAG_KEEP_VALUE_MAP) r.keep_value_map = true;
  }
  // developer knobs (like CFS_HIP_MAX_SLOTS): far threshold, HYB on / off
  if (const char *e = getenv("CFS_HIP_FAR_USES"))
    if (atoi(e) > 0) r.far_uses = atoi(e);
  if (const char *e = getenv("CFS_HIP_HYB")) r.hyb = atoi(e) != 0;
  if (const char *e = getenv("CFS_HIP_COST_MODEL")) r.cost_model = atoi(e) != 0;
  if (const char *e = getenv("CFS_HIP_COMBINE")) r.combine_siblings = atoi(e) != 0;
  if (const char *e = getenv("CFS_HIP_DETERMINISTIC")) r.deterministic = atoi(e) != 0;
  if (r.deterministic && r.block_threads == 256) r.block_threads = 512; // 512 / 1 024 threads
  if (o && (o->flags & CFS_HIP_FLAG_NO_HYB)) r.hyb = false;
  r.count_far = !r.hyb && !(o && (o->flags & (CFS_HIP_FLAG_NO_CALIBRATE | CFS_HIP_FLAG_NO_HYB)));
  // tuning knob for callers that cannot pass options (the C++ surface): LDS slots
  // per tile, like CFS_NUM_THREADS for the reference's partitions
  if (r.max_slots <= 0) {
    const char *e = getenv("CFS_HIP_MAX_SLOTS");
    if (e && atoi(e) > 0) r.max_slots = atoi(e);
  }
  // developer knob (tools/calib_probe.py): relative cost share of every persistent group
  if (const char *e = getenv("CFS_HIP_GROUP_SHARE_FILE"))
    if (FILE *f = fopen(e, "r")) {
      double v;
      while (fscanf(f, "%lf", &v) == 1) r.group_share.push_back(v);
      fclose(f);
    }
  return r;
}

// how many workgroups of the tile kernel are co-resident on a CU for the LDS
// budget the options ask for: the persistent grid is sized to exactly one
// resident wave of workgroups (a workgroup that has to wait for a slot would
// run as a second round and double the launch time)
template <typename V, int BLOCK> static int residency_one(size_t lds, int *nb, bool det = false) {
  const void *k = det ? (const void *)cfs_sym_tile_kernel<V, (BLOCK < 512 ? 512 : BLOCK), 0, true, true,
                                                          cfs_plan::kSlotsPerThread, true>
                      : (const void *)cfs_sym_tile_kernel<V, BLOCK, 0, true, true, cfs_plan::kSlotsPerThread>;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  int rc = raise_lds_limit(k, dev);
  if (rc) return rc;
  HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(nb, k, BLOCK, lds));
  return 0;
}
template <typename V> static int query_residency(cfs_plan::Options &po) {
  // the window the plan builder will allow (same rule: cfs_plan::chunk_layout)
  const cfs_plan::ChunkLayout L = cfs_plan::chunk_layout<V>(1 << 30, po);
  const int block = L.block;
  const int slot_bytes = cfs_plan::slot_lds_bytes<V>(po.deterministic);
  const size_t lds = (size_t)((L.max_slots + 63) / 64 * 64) * slot_bytes;
  int nb = 0, rc;
  switch (block) {
  case 256: rc = residency_one<V, 256>(lds, &nb); break;
  case 512: rc = residency_one<V, 512>(lds, &nb, po.deterministic); break;
  case 1024: rc = residency_one<V, 1024>(lds, &nb, po.deterministic); break;
  default: return 0; // build_plan reports the bad block size
  }
  if (rc) return rc;
  hipDeviceProp_t prop;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  HIPCHK(hipGetDeviceProperties(&prop, dev));
  po.wg_per_cu = nb > 0 ? nb : 1;
  po.num_cus = prop.multiProcessorCount;
  return 0;
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
  cfs_plan::PhaseTimer ct;
  int rc = ensure_init();
  if (rc) return rc;
  int cur_dev = 0;
  HIPCHK(hipGetDevice(&cur_dev));
  auto *m = new SymMatrix<V>();
  m->value_bytes = (int)sizeof(V);
  m->device = cur_dev;
  cfs_plan::Options po = to_opts(opt);
  if ((rc = query_residency<V>(po))) {
    delete m;
    return rc;
  }
  ct.lap("create: runtime + kernel residency");
  // kept until the window-shape step below has had its chance to reuse it
  cfs_plan::ScheduleSpace<V> space;
  const bool want_tuning = !(opt && (opt->flags & CFS_HIP_FLAG_NO_CALIBRATE));
  // the schedule: built on the GPU (cfs_devplan.hpp) when the options are covered, else -- and
  // whenever the device builder hands over -- by the host builder + upload
  const bool host_plan = (opt && (opt->flags & CFS_HIP_FLAG_HOST_PLAN)) ||
                         (getenv("CFS_HIP_DEVICE_PLAN") && atoi(getenv("CFS_HIP_DEVICE_PLAN")) == 0);
  auto build_handle = [&](SymMatrix<V> *h, cfs_plan::Options &o, cfs_plan::ScheduleSpace<V> *sp) -> int {
    h->nnz_caller = rowptr[n];
    if (!host_plan) {
      std::string why;
      const int r2 = cfs_dev::build<V>(n, rowptr, colind, values, nranks, rank,
                                       nranks > 1 ? row_splits : nullptr, o, *h, sp, why);
      if (r2 == 0) return h->adopt_device_schedule(); // (the builder's temporaries are gone by now)
      if (r2 < 0) return r2;
      if (getenv("CFS_PLAN_VERBOSE")) fprintf(stderr, "[cfs_hip] device builder hands over: %s\n", why.c_str());
      h->device_built = false;
      h->plan_note = why.empty() ? "handed over" : why;
    }
    if (!cfs_plan::build_plan<V>(n, rowptr, colind, values, nranks, rank, nranks > 1 ? row_splits : nullptr, o,
                                 h->P, sp))
      return set_err(plan_error_code(h->P.error), h->P.error);
    return h->upload();
  };
  rc = build_handle(m, po, want_tuning ? &space : nullptr);
  ct.lap("create: schedule (build + upload)");
  if (rc) {
    delete m;
    return rc;
  }
  // ---- window shape (Tuning::Aggressive; CFS_HIP_FLAG_NO_CALIBRATE skips it) -------
  // Default: 512 threads, 4 992 slots, two workgroups per CU.  A matrix that is
  // scheduled in clustered order (compact 3-D tiles) can be faster with ONE
  // workgroup of 1 024 threads and a window twice the size per CU -- half as many,
  // larger tiles, a quarter fewer halo slots, a shorter fold (Flan stand-in fp64:
  // 0.122 -> 0.116 ms per SpMV; fp32, ldoor, pwtk, Queen: no gain or a loss).
  // Both schedules are built and timed; the faster one is kept.
  const bool tuning = !(opt && (opt->flags & CFS_HIP_FLAG_NO_CALIBRATE)) &&
                      m->P.nnz_low >= (int64_t)2000000;
  // a measured step: build the alternative schedule, time ten SpMVs of each
  // (interleaved, best of three), keep the faster one
  DevBuf xb, yb;
  auto time_spmv = [&](SymMatrix<V> *h, float *ms) -> int {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
    int r2 = 0;
    for (int it = 0; it < 3 && !r2; it++) r2 = h->spmv_local(yb.p, xb.p, nullptr, (hipStream_t)0, 3);
    (void)hipEventRecord(a, (hipStream_t)0);
    for (int it = 0; it < 10 && !r2; it++) r2 = h->spmv_local(yb.p, xb.p, nullptr, (hipStream_t)0, 3);
    (void)hipEventRecord(b, (hipStream_t)0);
    if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(ms, a, b) != hipSuccess) r2 = -1;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return r2;
  };
  auto ensure_xy = [&]() -> int {
    if (xb.p) return 0;
    int r2;
    if ((r2 = xb.alloc((size_t)n * sizeof(V))) || (r2 = yb.alloc((size_t)m->rows() * sizeof(V)))) return r2;
    (void)hipMemset(xb.p, 0x3f, xb.bytes); // small positive values
    return 0;
  };
  // which instantiation of the tile kernel (sibling-combined atomics or plain) a handle
  // launches: same schedule, same arrays -- time ten SpMVs of each, three times
  auto choose_kernel = [&](SymMatrix<V> *h) -> int {
    if (!h->combine_ok || h->combine_forced) return 0;
    int r2 = ensure_xy();
    if (r2) return r2;
    const bool dflt = h->combine; // the size rule's choice (upload()): kept unless the other
    float t[2] = {1e30f, 1e30f};  // instantiation is clearly -- 3 %, best of three -- faster
    for (int round = 0; round < 3; round++)
      for (int c = 0; c < 2; c++) {
        float ms = 0;
        h->combine = c == 1;
        if (time_spmv(h, &ms)) {
          h->combine = dflt;
          return 0;
        }
        t[c] = std::min(t[c], ms);
      }
    h->combine = t[dflt ? 0 : 1] < 0.97f * t[dflt ? 1 : 0] ? !dflt : dflt;
    if (getenv("CFS_PLAN_VERBOSE"))
      fprintf(stderr, "[cfs_hip] tile kernel: plain %.1f us, sibling-combined atomics %.1f us\n", t[0] * 100.0,
              t[1] * 100.0);
    return 0;
  };
  auto try_alternative = [&](cfs_plan::Options &po2, cfs_plan::ScheduleSpace<V> *sp,
                             const char *what, int forced = -1) -> int {
    {
      int r2 = ensure_xy();
      if (r2) return r2;
    }
    auto *alt = new SymMatrix<V>();
    alt->value_bytes = (int)sizeof(V);
    alt->device = cur_dev;
    alt->nnz_caller = rowptr[n];
    float t_def = 0, t_alt = 0;
    bool ok = query_residency<V>(po2) == 0 && build_handle(alt, po2, sp) == 0 && choose_kernel(alt) == 0;
    for (int round = 0; ok && round < 3; round++) {
      float a = 0, b = 0;
      ok = time_spmv(m, &a) == 0 && time_spmv(alt, &b) == 0;
      t_def = round == 0 ? a : std::min(t_def, a);
      t_alt = round == 0 ? b : std::min(t_alt, b);
    }
    if (getenv("CFS_PLAN_VERBOSE"))
      fprintf(stderr, "[cfs_hip] %s: current %.1f us, alternative %.1f us%s\n", what, t_def * 100.0,
              t_alt * 100.0, ok ? "" : " (alternative not built)");
    // (CFS_HIP_KEEP_ALT=1: developer knob -- keep the alternative whatever the clock says,
    // so that each of the schedules tune() may end up with can be profiled on any box)
    const char *force = getenv("CFS_HIP_KEEP_ALT");
    if (force && atoi(force) != 0) forced = 1;
    if (ok && forced != 0 && (t_alt < 0.99f * t_def || forced == 1)) {
      delete m;
      m = alt;
      po = po2;
    } else {
      delete alt;
    }
    return 0;
  };
  if (tuning && (rc = choose_kernel(m))) {
    delete m;
    return rc;
  }
  // (natural-order schedules are tried too: a 1/8 row block of the Flan stand-in runs 11 %
  // faster with 256 workgroups of 1 024 threads -- a third less halo, half as many window
  // phases per CU -- while pwtk, ldoor and Queen stay with the default)
  if (tuning && po.block_threads == 0 && po.max_slots == 0) {
    cfs_plan::Options po2 = po;
    po2.block_threads = 1024;
    po2.max_slots = 2 * cfs_plan::kDefaultSlots;
    // (CFS_HIP_SHAPE=512|1024: developer knob -- the profiler passes of one evidence set
    // must run the schedule its bench line ran, whatever the clock says under the profiler)
    int forced = -1;
    if (const char *e = getenv("CFS_HIP_SHAPE")) forced = atoi(e) == 1024 ? 1 : (atoi(e) == 512 ? 0 : -1);
    if ((rc = try_alternative(po2, &space, "window shape 512 x 2 per CU vs 1024 x 1", forced))) {
      delete m;
      return rc;
    }
  }
  // ---- HYB by measurement (Format::sss, Tuning::Aggressive) --------------------------
  // A halo column that its tile uses once costs a slot, a slot-table entry, an x
  // gather, a strip store and a fold entry for one nonzero; as a FAR entry the nonzero
  // is stored twice and gathers x from L2.  Worth trying when such columns are a
  // noticeable share of the matrix (ldoor stand-in: 6 % of the stored nonzeros, halo
  // 2.55M -> 1.16M slots, 71 -> 66 us per SpMV; Flan, pwtk stand-ins: < 0.1 %, not tried).
  const bool may_hyb = !(opt && (opt->flags & (CFS_HIP_FLAG_NO_HYB | CFS_HIP_FLAG_HYB))) && !po.hyb;
  if (tuning && may_hyb && m->P.far_candidates * 33 >= m->P.nnz_low) {
    cfs_plan::Options po2 = po;
    po2.hyb = true;
    int forced = -1; // CFS_HIP_TAKE_HYB=1|0: keep / drop the alternative whatever the clock says (tests)
    if (const char *e = getenv("CFS_HIP_TAKE_HYB")) forced = atoi(e) != 0 ? 1 : 0;
    // (the kept upload / schedule space of the builds above serves this one too)
    if ((rc = try_alternative(po2, &space, "tile format vs HYB (far entries apart)", forced))) {
      delete m;
      return rc;
    }
  }
  space.drop(); // release the schedule-space matrix / the kept upload
  m->ablate_mode = opt ? (opt->flags & CFS_HIP_FLAG_ABLATE_MASK) : 0;
  *out = m;
  ct.lap("create: measured alternatives");
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


// ---------------------------------------------------------------------------
// native exchange (cfs_comm.hpp): RCCL over xGMI, or kernels / copies over peer access
// ---------------------------------------------------------------------------
int cfs_hip_comm_create(int ndev, const int *devices, int transport, cfs_hip_comm_t *out) {
  if (!out || ndev < 1 || ndev > cfs_rt::kMaxDevices) return set_err(CFS_HIP_ERR_ARG, "bad argument");
  *out = nullptr;
  int rc = ensure_init();
  if (rc) return rc;
  int visible = 0, cur = 0;
  HIPCHK(hipGetDeviceCount(&visible));
  HIPCHK(hipGetDevice(&cur));
  auto *c = new cfs_hip_comm_s();
  bool distinct = true;
  for (int g = 0; g < ndev; g++) {
    const int d = devices ? devices[g] : (cur + g) % std::max(1, visible);
    if (d < 0 || d >= visible) {
      delete c;
      return set_err(CFS_HIP_ERR_ARG, "bad device index");
    }
    for (int q : c->dev) distinct = distinct && q != d;
    c->dev.push_back(d);
  }
  cfs_comm::Rccl &R = cfs_comm::rccl();
  if (transport == CFS_HIP_TRANSPORT_RCCL && (!distinct || !R.ok)) {
    const std::string why = !distinct ? "RCCL needs one rank per device" : R.why;
    delete c;
    return set_err(CFS_HIP_ERR_UNSUPPORTED, "rccl transport: " + why);
  }
  c->use_rccl = transport != CFS_HIP_TRANSPORT_PEER && distinct && R.ok;
  if (!c->use_rccl) c->note = transport == CFS_HIP_TRANSPORT_PEER ? "asked for" : (!distinct ? "ranks share a device" : R.why);
  if (c->use_rccl) {
    c->comm.assign(ndev, nullptr);
    const int r2 = R.CommInitAll(c->comm.data(), ndev, c->dev.data());
    if (r2 != 0) {
      const std::string e = R.GetErrorString ? R.GetErrorString(r2) : "?";
      c->use_rccl = false; // (nothing to destroy)
      delete c;
      return set_err(CFS_HIP_ERR_DEVICE, "ncclCommInitAll: " + e);
    }
  } else {
    c->ptrs = std::vector<DevBuf>(ndev);
    c->ready.assign(ndev, nullptr);
    c->done.assign(ndev, nullptr);
    for (int g = 0; g < ndev; g++) {
      DeviceGuard dg(c->dev[g]);
      for (int q = 0; q < ndev; q++) // every rank reads every other rank's buffers
        if (c->dev[q] != c->dev[g]) {
          int can = 0;
          (void)hipDeviceCanAccessPeer(&can, c->dev[g], c->dev[q]);
          hipError_t e = can ? hipDeviceEnablePeerAccess(c->dev[q], 0) : hipErrorPeerAccessUnsupported;
          if (e == hipErrorPeerAccessAlreadyEnabled) {
            (void)hipGetLastError();
            e = hipSuccess;
          }
          if (e != hipSuccess) {
            delete c;
            return set_err(CFS_HIP_ERR_DEVICE, "peer access between the devices of the communicator is not available");
          }
        }
      if (hipEventCreateWithFlags(&c->ready[g], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&c->done[g], hipEventDisableTiming) != hipSuccess ||
          c->ptrs[g].alloc((size_t)ndev * sizeof(void *)) != 0) {
        delete c;
        return set_err(CFS_HIP_ERR_DEVICE, "event / table creation failed");
      }
    }
  }
  *out = c;
  return 0;
}
int cfs_hip_comm_info(cfs_hip_comm_t c, int *ndev, int *transport) {
  if (!c) return set_err(CFS_HIP_ERR_ARG, "null communicator");
  if (ndev) *ndev = (int)c->dev.size();
  if (transport) *transport = c->use_rccl ? CFS_HIP_TRANSPORT_RCCL : CFS_HIP_TRANSPORT_PEER;
  return 0;
}
int cfs_hip_comm_destroy(cfs_hip_comm_t c) {
  delete c;
  return 0;
}
// make `stream` (of rank `rank`) wait until the previous collective has consumed that rank's
// send buffer (peer transport: other ranks' kernels read it; RCCL orders on the stream itself)
int cfs_hip_comm_wait_consumed(cfs_hip_comm_t c, int rank, void *stream) {
  if (!c || rank < 0 || rank >= (int)c->dev.size()) return set_err(CFS_HIP_ERR_ARG, "bad argument");
  if (c->use_rccl || !c->done_valid) return 0;
  DeviceGuard dg(c->dev[rank]);
  for (size_t r = 0; r < c->dev.size(); r++) HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->done[r], 0));
  return 0;
}
int cfs_hip_comm_reduce_scatter(cfs_hip_comm_t c, void *const *send, void *const *recv, size_t count,
                                int value_bytes, void *const *streams) {
  if (!c || !send || !recv || !streams || (value_bytes != 4 && value_bytes != 8))
    return set_err(CFS_HIP_ERR_ARG, "bad argument");
  const int N = (int)c->dev.size();
  if (c->use_rccl) {
    cfs_comm::Rccl &R = cfs_comm::rccl();
    int r2 = R.GroupStart();
    for (int g = 0; g < N && r2 == 0; g++) {
      DeviceGuard dg(c->dev[g]);
      r2 = R.ReduceScatter(send[g], recv[g], count, value_bytes == 8 ? cfs_comm::kNcclFloat64 : cfs_comm::kNcclFloat32,
                           cfs_comm::kNcclSum, c->comm[g], (hipStream_t)streams[g]);
    }
    const int r3 = R.GroupEnd();
    if (r2 == 0) r2 = r3;
    if (r2 != 0) return set_err(CFS_HIP_ERR_DEVICE, std::string("ncclReduceScatter: ") + (R.GetErrorString ? R.GetErrorString(r2) : "?"));
    return 0;
  }
  // peer transport: rank r sums the r-th block of every rank's send buffer
  for (int g = 0; g < N; g++) {
    DeviceGuard dg(c->dev[g]);
    HIPCHK(hipEventRecord(c->ready[g], (hipStream_t)streams[g]));
  }
  for (int r = 0; r < N; r++) {
    DeviceGuard dg(c->dev[r]);
    hipStream_t st = (hipStream_t)streams[r];
    for (int g = 0; g < N; g++) HIPCHK(hipStreamWaitEvent(st, c->ready[g], 0));
    HIPCHK(hipMemcpyAsync(c->ptrs[r].p, send, (size_t)N * sizeof(void *), hipMemcpyHostToDevice, st));
    const int grid = (int)std::min<size_t>((count + 255) / 256, 2048);
    if (count) {
      if (value_bytes == 8)
        hipLaunchKernelGGL((cfs_comm::cfs_peer_sum_kernel<double>), dim3(grid), dim3(256), 0, st, (double *)recv[r],
                           (const double *const *)c->ptrs[r].p, N, (size_t)r * count, count);
      else
        hipLaunchKernelGGL((cfs_comm::cfs_peer_sum_kernel<float>), dim3(grid), dim3(256), 0, st, (float *)recv[r],
                           (const float *const *)c->ptrs[r].p, N, (size_t)r * count, count);
    }
    HIPCHK(hipEventRecord(c->done[r], st));
  }
  c->done_valid = true;
  HIPCHK(hipGetLastError());
  return 0;
}
int cfs_hip_comm_allgather(cfs_hip_comm_t c, void *const *send, void *const *recv, size_t count,
                           int value_bytes, void *const *streams) {
  if (!c || !send || !recv || !streams || (value_bytes != 4 && value_bytes != 8))
    return set_err(CFS_HIP_ERR_ARG, "bad argument");
  const int N = (int)c->dev.size();
  if (c->use_rccl) {
    cfs_comm::Rccl &R = cfs_comm::rccl();
    int r2 = R.GroupStart();
    for (int g = 0; g < N && r2 == 0; g++) {
      DeviceGuard dg(c->dev[g]);
      r2 = R.AllGather(send[g], recv[g], count, value_bytes == 8 ? cfs_comm::kNcclFloat64 : cfs_comm::kNcclFloat32,
                       c->comm[g], (hipStream_t)streams[g]);
    }
    const int r3 = R.GroupEnd();
    if (r2 == 0) r2 = r3;
    if (r2 != 0) return set_err(CFS_HIP_ERR_DEVICE, std::string("ncclAllGather: ") + (R.GetErrorString ? R.GetErrorString(r2) : "?"));
    return 0;
  }
  // peer transport: rank g pushes its block into every rank's receive buffer
  const size_t bytes = count * (size_t)value_bytes;
  for (int g = 0; g < N; g++) {
    DeviceGuard dg(c->dev[g]);
    hipStream_t st = (hipStream_t)streams[g];
    for (int r = 0; r < N && bytes; r++)
      HIPCHK(hipMemcpyPeerAsync((char *)recv[r] + (size_t)g * bytes, c->dev[r], send[g], c->dev[g], bytes, st));
    HIPCHK(hipEventRecord(c->ready[g], st));
  }
  for (int r = 0; r < N; r++) {
    DeviceGuard dg(c->dev[r]);
    for (int g = 0; g < N; g++) HIPCHK(hipStreamWaitEvent((hipStream_t)streams[r], c->ready[g], 0));
  }
  return 0;
}

// ---------------------------------------------------------------------------
// One host thread, N GPUs (the C++ surface with CFS_NUM_GPUS=N; reference knob:
// CFS_NUM_THREADS, src/runtime.cpp:10-21): N mirrored 1-D row-block shards, one per
// device, each on a stream of its own.  x and y stay where the caller put them (the
// handle's HOME device); a shard on another device reads x and writes its rows of y
// through peer access over xGMI.  An SpMV is ordered like any other work of the
// caller's stream: the shard streams wait for an event recorded on it, it waits for
// theirs.  (The performance path for several GPUs is one process per GPU, bench.py;
// this is the drop-in path of an unmodified single-process caller.)
// ---------------------------------------------------------------------------
struct MultiSym : cfs_hip_sym_s {
  std::vector<cfs_hip_sym_s *> shard;
  std::vector<int> dev, splits;
  std::vector<hipStream_t> st_;
  std::vector<hipEvent_t> done_;
  hipEvent_t start_ = nullptr;
  int n_ = 0;
  std::vector<int32_t> none_;
  // How a shard on another device than the handle's home reaches x and y:
  //   CFS_HIP_XMODE_REPLICATE (default)  x is REPLICATED (north-star / SURVEY 8e): one
  //       hipMemcpyPeerAsync home -> device per shard and SpMV into the shard's own copy,
  //       the kernels gather x and write their y block in LOCAL HBM, one peer copy brings
  //       the block home;
  //   CFS_HIP_XMODE_PEER  the kernels read x and write y in the home device's memory
  //       through peer access over xGMI (no copies, every gather crosses the fabric).
  // Shards on the home device itself never copy.  CFS_HIP_XMODE_REPLICATE_ALL copies for
  // every shard, home or not: the way a one-GPU box exercises the copy path.
  int xmode = CFS_HIP_XMODE_REPLICATE;
  std::vector<DevBuf> xrep, yloc; // per shard, on its device (allocated on first use)
  int ensure_copies(size_t g) {
    if (xrep.size() != shard.size()) {
      xrep = std::vector<DevBuf>(shard.size());
      yloc = std::vector<DevBuf>(shard.size());
    }
    if (xrep[g].p) return 0;
    DeviceGuard dg(dev[g]);
    int rc;
    if ((rc = xrep[g].alloc((size_t)n_ * value_bytes))) return rc;
    return yloc[g].alloc((size_t)(splits[g + 1] - splits[g]) * value_bytes);
  }
  bool copies(size_t g) const {
    return xmode == CFS_HIP_XMODE_REPLICATE_ALL || (xmode == CFS_HIP_XMODE_REPLICATE && dev[g] != device);
  }
  // Exchange form (CFS_HIP_FLAG_SHARD_EXCHANGE at create, or CFS_MULTI_EXCHANGE=reduce_scatter):
  // the shards keep their off-block entries two-sided, pack the contributions to rows of lower
  // ranks, scatter them into a dense vector of N equal blocks and ONE native reduce-scatter
  // (cfs_hip_comm_*: RCCL over xGMI, or the peer transport) hands every owner its sums --
  // the north-star's form, without Python.  The local fold runs beside the collective.
  cfs_hip_comm_s *comm = nullptr;
  int rs_rows = 0; // block length of the reduce-scatter (longest row block)
  std::vector<DevBuf> sbuf, pos, dense, rsout;
  std::vector<int> nsend_;
  template <typename V> int setup_exchange(int transport) {
    const int N = (int)shard.size();
    int rc = cfs_hip_comm_create(N, dev.data(), transport, &comm);
    if (rc) return rc;
    rs_rows = 0;
    for (int g = 0; g < N; g++) rs_rows = std::max(rs_rows, splits[g + 1] - splits[g]);
    sbuf = std::vector<DevBuf>(N);
    pos = std::vector<DevBuf>(N);
    dense = std::vector<DevBuf>(N);
    rsout = std::vector<DevBuf>(N);
    nsend_.assign(N, 0);
    for (int g = 0; g < N; g++) {
      DeviceGuard dg(dev[g]);
      const std::vector<int32_t> &rows = shard[g]->send_rows();
      nsend_[g] = (int)rows.size();
      std::vector<int32_t> p(rows.size());
      for (size_t k = 0; k < rows.size(); k++) {
        const int owner = (int)(std::upper_bound(splits.begin(), splits.end(), rows[k]) - splits.begin()) - 1;
        p[k] = owner * rs_rows + (rows[k] - splits[owner]);
      }
      if ((rc = pos[g].upload(p.data(), p.size() * 4)) || (rc = sbuf[g].alloc(std::max<size_t>(1, rows.size()) * sizeof(V))) ||
          (rc = dense[g].alloc((size_t)N * rs_rows * sizeof(V))) || (rc = rsout[g].alloc((size_t)rs_rows * sizeof(V))))
        return rc;
      // this shard receives nothing through the sparse route: recv side stays empty
    }
    return 0;
  }
  template <typename V> int spmv_exchange(void *y, const void *x, hipStream_t st) {
    const int N = (int)shard.size();
    HIPCHK(hipEventRecord(start_, st));
    std::vector<void *> sp(N), rp(N), streams(N);
    std::vector<const void *> xg(N);
    std::vector<void *> yg(N);
    for (int g = 0; g < N; g++) {
      DeviceGuard dg(dev[g]);
      HIPCHK(hipStreamWaitEvent(st_[g], start_, 0));
      int rc;
      xg[g] = x;
      yg[g] = (char *)y + (size_t)splits[g] * value_bytes;
      if (copies(g)) {
        if ((rc = ensure_copies(g))) return rc;
        HIPCHK(hipMemcpyPeerAsync(xrep[g].p, dev[g], x, device, (size_t)n_ * value_bytes, st_[g]));
        xg[g] = xrep[g].p;
        yg[g] = yloc[g].p;
      }
      if ((rc = cfs_hip_comm_wait_consumed(comm, g, st_[g]))) return rc;
      HIPCHK(hipMemsetAsync(dense[g].p, 0, (size_t)N * rs_rows * sizeof(V), st_[g]));
      rc = shard[g]->spmv_local(yg[g], xg[g], sbuf[g].p, st_[g], CFS_HIP_PHASE_TILES | CFS_HIP_PHASE_PACK);
      if (rc) return rc;
      if (nsend_[g] > 0)
        hipLaunchKernelGGL((cfs_scatter_pos_kernel<V>), dim3((nsend_[g] + 255) / 256), dim3(256), 0, st_[g],
                           (V *)dense[g].p, (const int32_t *)pos[g].p, (const V *)sbuf[g].p, nsend_[g]);
      sp[g] = dense[g].p;
      rp[g] = rsout[g].p;
      streams[g] = (void *)st_[g];
    }
    int rc = cfs_hip_comm_reduce_scatter(comm, sp.data(), rp.data(), (size_t)rs_rows, value_bytes, streams.data());
    if (rc) return rc;
    for (int g = 0; g < N; g++) {
      DeviceGuard dg(dev[g]);
      const int rows_g = splits[g + 1] - splits[g];
      // (stream order: the local fold is enqueued behind the collective of this rank; on the
      // RCCL transport the two run on the same stream, on the peer transport the sum kernel is
      // short -- overlapping them needs a second stream per shard and has not been measured)
      rc = shard[g]->spmv_local(yg[g], xg[g], sbuf[g].p, st_[g], CFS_HIP_PHASE_FOLD);
      if (rc) return rc;
      if (rows_g > 0)
        hipLaunchKernelGGL((cfs_add_rows_kernel<V>), dim3((rows_g + 255) / 256), dim3(256), 0, st_[g], (V *)yg[g],
                           (const V *)rsout[g].p, rows_g);
      if (copies(g) && rows_g > 0)
        HIPCHK(hipMemcpyPeerAsync((char *)y + (size_t)splits[g] * value_bytes, device, yloc[g].p, dev[g],
                                  (size_t)rows_g * value_bytes, st_[g]));
      HIPCHK(hipEventRecord(done_[g], st_[g]));
    }
    for (int g = 0; g < N; g++) HIPCHK(hipStreamWaitEvent(st, done_[g], 0));
    HIPCHK(hipGetLastError());
    return 0;
  }
  ~MultiSym() override {
    for (size_t g = 0; g < shard.size(); g++) {
      DeviceGuard dg(dev[g]);
      if (g < st_.size() && st_[g]) (void)hipStreamSynchronize(st_[g]);
      delete shard[g];
      if (g < done_.size() && done_[g]) (void)hipEventDestroy(done_[g]);
      if (g < st_.size() && st_[g]) (void)hipStreamDestroy(st_[g]);
    }
    if (start_) {
      DeviceGuard dg(device);
      (void)hipEventDestroy(start_);
    }
    delete comm;
  }
  int spmv_local(void *y, const void *x, void *, hipStream_t st, int phases) override {
    if (comm) return value_bytes == 8 ? spmv_exchange<double>(y, x, st) : spmv_exchange<float>(y, x, st);
    HIPCHK(hipEventRecord(start_, st));
    for (size_t g = 0; g < shard.size(); g++) {
      DeviceGuard dg(dev[g]);
      HIPCHK(hipStreamWaitEvent(st_[g], start_, 0));
      char *yg = (char *)y + (size_t)splits[g] * value_bytes;
      const size_t ybytes = (size_t)(splits[g + 1] - splits[g]) * value_bytes;
      int rc;
      if (copies(g)) { // replicated x, local y block
        if ((rc = ensure_copies(g))) return rc;
        HIPCHK(hipMemcpyPeerAsync(xrep[g].p, dev[g], x, device, (size_t)n_ * value_bytes, st_[g]));
        rc = shard[g]->spmv_local(yloc[g].p, xrep[g].p, nullptr, st_[g],
                                  phases & (CFS_HIP_PHASE_TILES | CFS_HIP_PHASE_FOLD));
        if (rc) return rc;
        if (ybytes) HIPCHK(hipMemcpyPeerAsync(yg, device, yloc[g].p, dev[g], ybytes, st_[g]));
      } else {
        rc = shard[g]->spmv_local(yg, x, nullptr, st_[g], phases & (CFS_HIP_PHASE_TILES | CFS_HIP_PHASE_FOLD));
        if (rc) return rc;
      }
      HIPCHK(hipEventRecord(done_[g], st_[g]));
    }
    for (size_t g = 0; g < shard.size(); g++) HIPCHK(hipStreamWaitEvent(st, done_[g], 0));
    return 0;
  }
  int recv_fold(void *, const void *, hipStream_t) override { return 0; }
  int set_recv(int, const int *) override { return set_err(CFS_HIP_ERR_ARG, "not a shard"); }
  void stats(cfs_hip_sym_stats *o) override {
    memset(o, 0, sizeof *o);
    for (auto *h : shard) {
      cfs_hip_sym_stats t;
      h->stats(&t);
      o->nnz_low += t.nnz_low;
      o->nnz_diag += t.nnz_diag;
      o->nnz_full += t.nnz_full;
      o->ntiles += t.ntiles;
      o->nslices += t.nslices;
      o->halo_slots += t.halo_slots;
      o->fold_rows += t.fold_rows;
      o->bytes_algorithmic += t.bytes_algorithmic;
      o->bytes_streamed += t.bytes_streamed;
      o->device_bytes += t.device_bytes;
      o->mirror_entries += t.mirror_entries;
      o->far_entries += t.far_entries;
      o->ngroups += t.ngroups;
      o->max_slots_used = std::max(o->max_slots_used, t.max_slots_used);
      o->lds_bytes = std::max(o->lds_bytes, t.lds_bytes);
      o->block_threads = t.block_threads;
      o->value_bytes = t.value_bytes;
    }
    o->n = n_;
    o->row_begin = 0;
    o->row_end = n_;
  }
  const std::vector<int32_t> &send_counts() override { return none_; }
  const std::vector<int32_t> &send_rows() override { return none_; }
  int n() override { return n_; }
  int rows() override { return n_; }
  int timeline(void *, const void *, unsigned long long *, int, int *) override {
    return set_err(CFS_HIP_ERR_ARG, "no timeline for a multi-device handle");
  }
  int group_features(long long *, int, int *) override {
    return set_err(CFS_HIP_ERR_ARG, "no group features for a multi-device handle");
  }
  int update_values(const void *values_dev, long long nnz, hipStream_t st) override {
    // (values_dev lives on the home device; shards on other devices read it over peer access)
    HIPCHK(hipStreamSynchronize(st));
    for (size_t g = 0; g < shard.size(); g++) {
      DeviceGuard dg(dev[g]);
      int rc = shard[g]->update_values(values_dev, nnz, st_[g]);
      if (rc) return rc;
      HIPCHK(hipStreamSynchronize(st_[g]));
    }
    return 0;
  }
  int ngpus() const { return (int)shard.size(); }
};

template <typename V>
static int sym_create_multi(int n, const int *rowptr, const int *colind, const V *values, int ngpus,
                            const int *devices, const cfs_hip_options *opt, cfs_hip_sym_t *out) {
  if (!out) return set_err(CFS_HIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (ngpus < 1 || ngpus > cfs_rt::kMaxDevices) return set_err(CFS_HIP_ERR_ARG, "bad ngpus");
  if (n < 0 || !rowptr) return set_err(CFS_HIP_ERR_ARG, "null CSR array");
  int rc = ensure_init();
  if (rc) return rc;
  int home = 0, ndev = 0;
  HIPCHK(hipGetDevice(&home));
  HIPCHK(hipGetDeviceCount(&ndev));
  auto *m = new MultiSym();
  m->value_bytes = (int)sizeof(V);
  m->device = home;
  m->n_ = n;
  m->splits.assign(ngpus + 1, 0);
  cfs_plan::balanced_splits(n, rowptr, colind, ngpus, m->splits.data());
  if (const char *e = getenv("CFS_MULTI_X")) // peer | replicate | replicate_all
    m->xmode = !strcmp(e, "peer") ? CFS_HIP_XMODE_PEER
               : !strcmp(e, "replicate_all") ? CFS_HIP_XMODE_REPLICATE_ALL : CFS_HIP_XMODE_REPLICATE;
  {
    DeviceGuard dg(home);
    if (hipEventCreateWithFlags(&m->start_, hipEventDisableTiming) != hipSuccess) {
      delete m;
      return set_err(CFS_HIP_ERR_DEVICE, "hipEventCreate failed");
    }
  }
  cfs_hip_options o2;
  memset(&o2, 0, sizeof o2);
  if (opt) o2 = *opt;
  // default: mirrored shards, nothing to exchange.  With CFS_HIP_FLAG_SHARD_EXCHANGE (or
  // CFS_MULTI_EXCHANGE=reduce_scatter) the shards take the exchange form and one native
  // reduce-scatter per SpMV (MultiSym::spmv_exchange)
  bool exchange = (o2.flags & CFS_HIP_FLAG_SHARD_EXCHANGE) != 0;
  if (const char *e = getenv("CFS_MULTI_EXCHANGE")) exchange = !strcmp(e, "reduce_scatter");
  if (ngpus < 2) exchange = false;
  if (exchange) o2.flags = (o2.flags | CFS_HIP_FLAG_SHARD_EXCHANGE) & ~(CFS_HIP_FLAG_HYB);
  else o2.flags &= ~CFS_HIP_FLAG_SHARD_EXCHANGE;
  for (int g = 0; g < ngpus; g++) {
    // devices[g] when given, else the visible devices round-robin (several shards may
    // share a device: that is how a one-GPU box rehearses the path)
    const int d = devices ? devices[g] : (home + g) % std::max(1, ndev);
    if (d < 0 || d >= ndev) {
      delete m;
      return set_err(CFS_HIP_ERR_ARG, "bad device index");
    }
    DeviceGuard dg(d);
    if (d != home) { // the shard reads x / writes y on the home device
      int can = 0;
      (void)hipDeviceCanAccessPeer(&can, d, home);
      hipError_t e = can ? hipDeviceEnablePeerAccess(home, 0) : hipErrorPeerAccessUnsupported;
      if (e == hipErrorPeerAccessAlreadyEnabled) {
        (void)hipGetLastError();
        e = hipSuccess;
      }
      if (e != hipSuccess) {
        delete m;
        return set_err(CFS_HIP_ERR_DEVICE, "device " + std::to_string(d) + " cannot access device " +
                                               std::to_string(home) + " (peer access)");
      }
    }
    hipStream_t st = nullptr;
    hipEvent_t ev = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      delete m;
      return set_err(CFS_HIP_ERR_DEVICE, "stream / event creation failed");
    }
    cfs_hip_sym_t h = nullptr;
    rc = sym_create<V>(n, rowptr, colind, values, ngpus, g, m->splits.data(), &o2, &h);
    m->dev.push_back(d);
    m->st_.push_back(st);
    m->done_.push_back(ev);
    if (rc) {
      std::string e = cfs_rt::last_error();
      m->shard.push_back(nullptr);
      delete m;
      return set_err(rc, e);
    }
    m->shard.push_back(h);
  }
  if (exchange) {
    int transport = CFS_HIP_TRANSPORT_AUTO;
    if (const char *e = getenv("CFS_MULTI_TRANSPORT"))
      transport = !strcmp(e, "rccl") ? CFS_HIP_TRANSPORT_RCCL : (!strcmp(e, "peer") ? CFS_HIP_TRANSPORT_PEER : CFS_HIP_TRANSPORT_AUTO);
    if ((rc = m->template setup_exchange<V>(transport))) {
      std::string e = cfs_rt::last_error();
      delete m;
      return set_err(rc, e);
    }
  }
  *out = m;
  return 0;
}
int cfs_hip_sym_create_multi_f64(int n, const int *rowptr, const int *colind, const double *values,
                                 int ngpus, const int *devices, const cfs_hip_options *opt,
                                 cfs_hip_sym_t *out) {
  return sym_create_multi<double>(n, rowptr, colind, values, ngpus, devices, opt, out);
}
int cfs_hip_sym_create_multi_f32(int n, const int *rowptr, const int *colind, const float *values,
                                 int ngpus, const int *devices, const cfs_hip_options *opt,
                                 cfs_hip_sym_t *out) {
  return sym_create_multi<float>(n, rowptr, colind, values, ngpus, devices, opt, out);
}
int cfs_hip_sym_multi_set_xmode(cfs_hip_sym_t h, int xmode) {
  auto *m = dynamic_cast<MultiSym *>(h);
  if (!m) return set_err(CFS_HIP_ERR_ARG, "not a multi-device handle");
  if (xmode != CFS_HIP_XMODE_PEER && xmode != CFS_HIP_XMODE_REPLICATE && xmode != CFS_HIP_XMODE_REPLICATE_ALL)
    return set_err(CFS_HIP_ERR_ARG, "unknown x mode");
  // pending SpMVs of the other mode finish first
  for (size_t g = 0; g < m->shard.size(); g++) {
    DeviceGuard dg(m->dev[g]);
    HIPCHK(hipStreamSynchronize(m->st_[g]));
  }
  m->xmode = xmode;
  return 0;
}
int cfs_hip_sym_multi_devices(cfs_hip_sym_t h, int *devices, int capacity, int *distinct) {
  auto *m = dynamic_cast<MultiSym *>(h);
  if (!m || !distinct) return set_err(CFS_HIP_ERR_ARG, "not a multi-device handle");
  std::vector<int> seen;
  for (size_t g = 0; g < m->dev.size(); g++) {
    if (devices && (int)g < capacity) devices[g] = m->dev[g];
    if (std::find(seen.begin(), seen.end(), m->dev[g]) == seen.end()) seen.push_back(m->dev[g]);
  }
  *distinct = (int)seen.size();
  return 0;
}
int cfs_hip_sym_num_gpus(cfs_hip_sym_t h, int *ngpus) {
  if (!h || !ngpus) return set_err(CFS_HIP_ERR_ARG, "null argument");
  auto *m = dynamic_cast<MultiSym *>(h);
  *ngpus = m ? m->ngpus() : 1;
  return 0;
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

// the async entry points take device pointers on the caller's stream; the
// placement of a (y, x) pair is checked once (hipPointerGetAttributes costs more
// than the launch): vectors on another device than the matrix would fault or, with
// peer access, silently run over the fabric
static int check_placement(cfs_hip_sym_t h, const void *y, const void *x) {
  if (h->ok_x == x && h->ok_y == y) return 0;
  const cfs_rt::PtrInfo xi = cfs_rt::classify(x), yi = cfs_rt::classify(y);
  if (!xi.device || !yi.device)
    return set_err(CFS_HIP_ERR_ARG, "async entry points need device pointers (use cfs_hip_sym_spmv)");
  if (xi.dev != h->device || yi.dev != h->device)
    return set_err(CFS_HIP_ERR_ARG, "x / y live on device " +
                                        std::to_string(xi.dev != h->device ? xi.dev : yi.dev) +
                                        ", the matrix on device " + std::to_string(h->device));
  h->ok_x = x;
  h->ok_y = y;
  return 0;
}

int cfs_hip_sym_spmv_async(cfs_hip_sym_t h, void *y, const void *x, void *stream) {
  if (!h || !y || !x) return set_err(CFS_HIP_ERR_ARG, "null argument");
  if (!h->send_rows().empty())
    return set_err(CFS_HIP_ERR_ARG, "sharded handle: use cfs_hip_sym_spmv_local_async");
  int rc = check_placement(h, y, x);
  if (rc) return rc;
  DeviceGuard g(h->device);
  return h->spmv_local(y, x, nullptr, (hipStream_t)stream);
}

int cfs_hip_sym_cg(cfs_hip_sym_t h, void *u_dev, const void *b_dev, double tol, int maxiter, int check_every,
                   int *iterations, double *relres, void *stream) {
  if (!h || !u_dev || !b_dev) return set_err(CFS_HIP_ERR_ARG, "null argument");
  if (u_dev == b_dev) return set_err(CFS_HIP_ERR_ARG, "cg: u and b must be different vectors");
  if (iterations) *iterations = 0;
  if (relres) *relres = 0.0;
  if (!h->send_rows().empty())
    return set_err(CFS_HIP_ERR_UNSUPPORTED, "cg: a handle of the whole matrix (a loop over shards of several processes: cfs_spmv_amd/solver.py)");
  int rc = check_placement(h, u_dev, b_dev);
  if (rc) return rc;
  h->ok_x = h->ok_y = nullptr; // (the iteration's own vectors are library memory on the handle's device)
  DeviceGuard g(h->device);
  if (h->value_bytes == 8)
    return cfs_solver::cg<double>(h, u_dev, b_dev, tol, maxiter, check_every, iterations, relres, (hipStream_t)stream);
  return cfs_solver::cg<float>(h, u_dev, b_dev, tol, maxiter, check_every, iterations, relres, (hipStream_t)stream);
}

int cfs_hip_sym_spmv(cfs_hip_sym_t h, void *y, const void *x) {
  if (!h || !y || !x) return set_err(CFS_HIP_ERR_ARG, "null argument");
  if (!h->send_rows().empty())
    return set_err(CFS_HIP_ERR_ARG, "sharded handle: use cfs_hip_sym_spmv_local_async");
  const size_t vb = (size_t)h->value_bytes;
  return run_staged(h->device, h->stage, (size_t)h->n() * vb, (size_t)h->rows() * vb, y, x,
                    [&](void *yd, const void *xd, hipStream_t st) {
                      return h->spmv_local(yd, xd, nullptr, st);
                    });
}

// New numeric values, same sparsity pattern (a matrix reassembled in every step of a
// non-linear solve): the caller's full CSR value array -- host or device pointer -- is
// poured into the existing schedule by a device kernel; tune() is not repeated.
static int update_values(cfs_hip_sym_t h, const void *values, long long nnz, size_t vb) {
  if (!h || !values) return set_err(CFS_HIP_ERR_ARG, "null argument");
  if ((size_t)h->value_bytes != vb) return set_err(CFS_HIP_ERR_ARG, "value type differs from the handle's");
  cfs_rt::DevCtx *ctx;
  int rc = cfs_rt::device_ctx(h->device, &ctx);
  if (rc) return rc;
  DeviceGuard g(h->device);
  const cfs_rt::PtrInfo vi = cfs_rt::classify(values);
  DevBuf tmp;
  const void *src = values;
  if (!vi.device) {
    if ((rc = tmp.alloc((size_t)nnz * vb))) return rc;
    HIPCHK(hipMemcpyAsync(tmp.p, values, (size_t)nnz * vb, hipMemcpyHostToDevice, ctx->stream));
    src = tmp.p;
  } else if (vi.dev != h->device) {
    return set_err(CFS_HIP_ERR_ARG, "values live on another device than the matrix");
  }
  if ((rc = h->update_values(src, nnz, ctx->stream))) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream)); // tmp goes away; later SpMVs on any stream see the values
  return 0;
}
int cfs_hip_sym_update_values_f64(cfs_hip_sym_t h, const double *values, long long nnz) {
  return update_values(h, values, nnz, 8);
}
int cfs_hip_sym_update_values_f32(cfs_hip_sym_t h, const float *values, long long nnz) {
  return update_values(h, values, nnz, 4);
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
  int rc = check_placement(h, y, x);
  if (rc) return rc;
  DeviceGuard g(h->device);
  return h->spmv_local(y, x, send, (hipStream_t)stream);
}
int cfs_hip_sym_spmv_phases_async(cfs_hip_sym_t h, void *y, const void *x, void *send,
                                  int phases, void *stream) {
  if (!h || !y || !x) return set_err(CFS_HIP_ERR_ARG, "null argument");
  int rc = check_placement(h, y, x);
  if (rc) return rc;
  DeviceGuard g(h->device);
  return h->spmv_local(y, x, send, (hipStream_t)stream, phases);
}
int cfs_hip_sym_recv_fold_async(cfs_hip_sym_t h, void *y, const void *recv, void *stream) {
  if (!h || !y) return set_err(CFS_HIP_ERR_ARG, "null argument");
  DeviceGuard g(h->device);
  return h->recv_fold(y, recv, (hipStream_t)stream);
}

int cfs_hip_sym_debug_timeline(cfs_hip_sym_t h, void *y_dev, const void *x_dev,
                               unsigned long long *stamps, int capacity_words, int *ngroups) {
  if (!h || !y_dev || !x_dev || !stamps || !ngroups) return set_err(CFS_HIP_ERR_ARG, "null argument");
  return h->timeline(y_dev, x_dev, stamps, capacity_words, ngroups);
}

int cfs_hip_sym_debug_group_features(cfs_hip_sym_t h, long long *out, int capacity_words,
                                     int *ngroups) {
  if (!h || !out || !ngroups) return set_err(CFS_HIP_ERR_ARG, "null argument");
  return h->group_features(out, capacity_words, ngroups);
}

int cfs_hip_sym_get_stats(cfs_hip_sym_t h, cfs_hip_sym_stats *out) {
  if (!h || !out) return set_err(CFS_HIP_ERR_ARG, "null argument");
  h->stats(out);
  return 0;
}

// developer / test: digests of the schedule's device arrays (cfs_hip.h)
static unsigned long long fnv1a(const void *p, size_t n, unsigned long long h = 1469598103934665603ull) {
  const unsigned char *b = (const unsigned char *)p;
  for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}
template <typename V> static int sym_digest(SymMatrix<V> *m, unsigned long long *w) {
  DeviceGuard g(m->device);
  HIPCHK(hipDeviceSynchronize());
  std::vector<unsigned char> buf;
  auto dig = [&](const DevBuf &b, size_t bytes, unsigned long long *out) -> int {
    *out = 0;
    if (!b.p || bytes == 0) return 0;
    buf.resize(bytes);
    HIPCHK(hipMemcpy(buf.data(), b.p, bytes, hipMemcpyDeviceToHost));
    *out = fnv1a(buf.data(), bytes);
    return 0;
  };
  const SymPlan<V> &P = m->P;
  const size_t T = P.tiles.size(), G = P.group_first.size(), s = sizeof(V);
  int64_t nsl = 0, nvr = 0;
  for (const Tile &t : P.tiles) nsl += t.nslots, nvr += t.nvrows;
  int rc, k = 0;
  { // tiles, aexp masked (the device builder does not compute it: the deterministic build only)
    std::vector<Tile> tt(T);
    if (T) HIPCHK(hipMemcpy(tt.data(), m->tiles.p, T * sizeof(Tile), hipMemcpyDeviceToHost));
    for (auto &t : tt) t.aexp = 0;
    w[k++] = fnv1a(tt.data(), T * sizeof(Tile));
    std::vector<Tile> gf(G);
    if (G) HIPCHK(hipMemcpy(gf.data(), m->gfirst.p, G * sizeof(Tile), hipMemcpyDeviceToHost));
    for (auto &t : gf) t.aexp = 0;
    w[k++] = fnv1a(gf.data(), G * sizeof(Tile));
  }
  if ((rc = dig(m->group_ptr, G * sizeof(int2), &w[k++]))) return rc;
  if ((rc = dig(m->slot_col, (size_t)nsl * 4, &w[k++]))) return rc;
  if ((rc = dig(m->rowinfo, (size_t)nvr * 4, &w[k++]))) return rc;
  if ((rc = dig(m->diag, (size_t)nvr * s, &w[k++]))) return rc;
  if ((rc = dig(m->slice_meta, (size_t)m->nslices * 16, &w[k++]))) return rc;
  if ((rc = dig(m->leadlane, (size_t)m->nslices * 64, &w[k++]))) return rc;
  if ((rc = dig(m->vals, (size_t)m->stream_len * s, &w[k++]))) return rc;
  if ((rc = dig(m->slots, (size_t)m->slot_len * 2, &w[k++]))) return rc;
  if ((rc = dig(m->cvals, (size_t)m->coo_len * s, &w[k++]))) return rc;
  if ((rc = dig(m->crows, (size_t)m->coo_len * 2, &w[k++]))) return rc;
  if ((rc = dig(m->ccols, (size_t)m->coo_len * 2, &w[k++]))) return rc;
  if ((rc = dig(m->fold_rec, (size_t)(m->nfold + 1) * sizeof(int4), &w[k++]))) return rc;
  if ((rc = dig(m->fold_idx, m->fold_idx.bytes, &w[k++]))) return rc;
  if ((rc = dig(m->val_map, m->has_value_map ? (size_t)m->stream_len * 4 : 0, &w[k++]))) return rc;
  if ((rc = dig(m->cval_map, m->has_value_map ? (size_t)m->coo_len * 4 : 0, &w[k++]))) return rc;
  if ((rc = dig(m->diag_map, m->has_value_map ? (size_t)nvr * 4 : 0, &w[k++]))) return rc;
  w[k++] = (unsigned long long)m->P.lds_slots | ((unsigned long long)m->P.ngroups << 32);
  if ((rc = dig(m->slot_exp, m->P.deterministic ? (size_t)nsl * 2 : 0, &w[k++]))) return rc;
  if ((rc = dig(m->send_ptr, m->nsend > 0 ? ((size_t)m->nsend + 1) * 4 : 0, &w[k++]))) return rc;
  if ((rc = dig(m->send_idx, m->nsend > 0 ? (size_t)m->P.send_ptr.back() * 4 : 0, &w[k++]))) return rc;
  { // far sections (Format::hyb)
    const size_t fl = (size_t)P.far_len;
    if ((rc = dig(m->fvals, fl * s, &w[k++]))) return rc;
    if ((rc = dig(m->frows, fl * 2, &w[k++]))) return rc;
    if ((rc = dig(m->fcols, fl * 4, &w[k++]))) return rc;
    if ((rc = dig(m->fval_map, m->has_value_map ? fl * 4 : 0, &w[k++]))) return rc;
    w[k++] = (unsigned long long)P.far_entries;
  }
  w[CFS_HIP_DIGEST_WORDS - 1] = m->device_built ? 1ull : 0ull;
  return 0;
}
int cfs_hip_sym_debug_plan_note(cfs_hip_sym_t h, char *buf, int capacity) {
  if (!h || !buf || capacity < 1) return set_err(CFS_HIP_ERR_ARG, "bad argument");
  snprintf(buf, (size_t)capacity, "%s", h->plan_note.c_str());
  return 0;
}
int cfs_hip_sym_debug_digest(cfs_hip_sym_t h, unsigned long long *words, int capacity_words) {
  if (!h || !words || capacity_words < CFS_HIP_DIGEST_WORDS) return set_err(CFS_HIP_ERR_ARG, "bad argument");
  for (int i = 0; i < CFS_HIP_DIGEST_WORDS; i++) words[i] = 0;
  if (auto *d = dynamic_cast<SymMatrix<double> *>(h)) return sym_digest<double>(d, words);
  if (auto *f = dynamic_cast<SymMatrix<float> *>(h)) return sym_digest<float>(f, words);
  return set_err(CFS_HIP_ERR_ARG, "no digest for a multi-device handle");
}

// ---- host-only plan self-check (no device needed) -------------------------
template <typename V>
static int plan_check(int n, const int *rowptr, const int *colind, const V *values,
                      int nranks, int rank, const int *row_splits,
                      const cfs_hip_options *opt, cfs_hip_plan_report *rep) {
  if (!rep) return set_err(CFS_HIP_ERR_ARG, "report is NULL");
  memset(rep, 0, sizeof *rep);
  if (n < 0 || !rowptr || (n > 0 && rowptr[n] > 0 && (!colind || !values)))
    return set_err(CFS_HIP_ERR_ARG, "null CSR array");
  if (nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !row_splits))
    return set_err(CFS_HIP_ERR_ARG, "bad rank / nranks / row_splits");
  SymPlan<V> P;
  if (!cfs_plan::build_plan<V>(n, rowptr, colind, values, nranks, rank,
                               nranks > 1 ? row_splits : nullptr, to_opts(opt), P))
    return set_err(plan_error_code(P.error), P.error);
  std::vector<int32_t> r, c, fr, fc, ur, uc;
  std::vector<V> v, fv, uv;
  cfs_plan::decode_plan(P, r, c, v, &fr, &fc, &fv, &ur, &uc, &uv);
  rep->ntiles = (int)P.tiles.size();
  rep->ngroups = P.ngroups;
  rep->nslices = (int64_t)P.slice_meta.size();
  rep->halo_slots = P.nhalo;
  rep->stream_len = P.stream_len;
  rep->nnz_low = P.nnz_low;
  rep->lds_slots = P.lds_slots;
  rep->fold_rows = (int64_t)P.fold_row.size();
  rep->remote_vals = (int64_t)P.send_row.size();
  rep->decoded = (int64_t)r.size();
  rep->far_entries = P.far_entries;
  // (1) decoded triples == strict lower triangle of the owned rows, as
  // multisets of (row, col, value bits): clustering moves an entry to the row
  // of its later end and back, and reorders entries inside rows
  int64_t bad = 0;
  {
    struct Tr {
      int32_t r, c;
      uint64_t v;
      bool operator<(const Tr &o) const {
        return r != o.r ? r < o.r : (c != o.c ? c < o.c : v < o.v);
      }
      bool operator!=(const Tr &o) const { return r != o.r || c != o.c || v != o.v; }
    };
    auto bits = [](V x) {
      uint64_t u = 0;
      memcpy(&u, &x, sizeof(V));
      return u;
    };
    { // HYB: the mirrored far entries are exactly the far entries of the lower side
      std::vector<Tr> F, U;
      for (size_t k = 0; k < fr.size(); k++) F.push_back(Tr{fr[k], fc[k], bits(fv[k])});
      for (size_t k = 0; k < ur.size(); k++) U.push_back(Tr{ur[k], uc[k], bits(uv[k])});
      std::sort(F.begin(), F.end());
      std::sort(U.begin(), U.end());
      if (F.size() != U.size() || (int64_t)F.size() != P.far_entries) bad++;
      for (size_t k = 0; k < std::min(F.size(), U.size()); k++)
        if (F[k] != U[k]) bad++;
    }
    std::vector<Tr> A, B;
    A.reserve(r.size());
    B.reserve(r.size());
    for (size_t k = 0; k < r.size(); k++) A.push_back(Tr{r[k], c[k], bits(v[k])});
    auto lower_pos = [&](int hi, int lo) {
      for (int q = rowptr[hi]; q < rowptr[hi + 1]; q++)
        if (colind[q] == lo) return q;
      return -1;
    };
    int64_t mirrored = 0;
    for (int i = P.row_begin; i < P.row_end; i++)
      for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
        if (colind[j] < i) B.push_back(Tr{i, colind[j], bits(values[j])});
        else if (P.mirrored && colind[j] >= P.row_end) {
          // mirrored shard: the entry (c, i) of a higher rank's row c, with the
          // value of the LOWER triangle
          const int q = lower_pos(colind[j], i);
          if (q < 0) bad++;
          else B.push_back(Tr{colind[j], i, bits(values[q])});
          mirrored++;
        }
      }
    if (mirrored != P.mirror_entries) bad++;
    rep->mirror_entries = P.mirror_entries;
    // y window classes: slots [nown, ny) are in-block columns, [ny, nslots) are not
    for (const Tile &t : P.tiles) {
      if (t.ny < t.nown || t.ny > t.nslots) bad++;
      for (int h = 0; h < t.nslots - t.nown; h++) {
        const int c = P.halo_col[t.halo_off + h];
        const bool inblock = c >= P.row_begin && c < P.row_end;
        // mirrored shard: exactly the in-block columns have a y window entry;
        // exchange form / whole matrix: every slot has one, no column right of the block
        if (P.mirrored ? inblock != (t.nown + h < t.ny) : (t.ny != t.nslots || c >= P.row_end)) bad++;
      }
    }
    if (P.mirrored && !P.send_row.empty()) bad++;
    std::sort(A.begin(), A.end());
    std::sort(B.begin(), B.end());
    if (A.size() != B.size()) bad += 1 + (int64_t)(A.size() > B.size() ? A.size() - B.size() : B.size() - A.size());
    for (size_t k = 0; k < std::min(A.size(), B.size()); k++)
      if (A[k] != B[k]) bad++;
    // every own row appears exactly once in the slot table, at its diagonal slot
    std::vector<char> seen_row(P.row_end - P.row_begin, 0);
    for (const Tile &t : P.tiles)
      for (int i = 0; i < t.nown; i++) {
        int o = P.slot_col[t.slot_off + i] - P.row_begin;
        if (o < 0 || o >= P.row_end - P.row_begin || seen_row[o]++) bad++;
      }
    for (char ch : seen_row)
      if (!ch) bad++;
  }
  // (1b) CFS_HIP_FLAG_KEEP_VALUE_MAP: every stored value is the caller's value at its
  // recorded position, and every stored entry has a position
  if (!P.val_map.empty()) {
    auto same = [&](const auto &arr, const auto &map, int64_t *mapped) {
      for (size_t k = 0; k < map.size() && k < arr.size(); k++) {
        if (map[k] < 0) continue;
        (*mapped)++;
        if (map[k] >= rowptr[n] || memcmp(&arr[k], &values[map[k]], sizeof(V)) != 0) bad++;
      }
    };
    int64_t mv = 0, mc = 0, mf = 0, md = 0;
    same(P.vals, P.val_map, &mv);
    same(P.cvals, P.cval_map, &mc);
    same(P.fvals, P.fval_map, &mf);
    same(P.diag, P.diag_map, &md);
    if (mv + mc != P.nnz_low + P.mirror_entries - P.far_entries || mf != 2 * P.far_entries) bad++;
    if (md > (int64_t)(P.row_end - P.row_begin)) bad++;
  }
  // (1c) sibling chains (leadlane bits 6 / 7), what entry_combined relies on: a lane that
  // hands its products to its left neighbour is a follower, shares that neighbour's group
  // and row of 16 lanes, has no more packets than it; runs hold at most three lanes; bit 7
  // of a lane mirrors bit 6 of the next one
  for (const Tile &t : P.tiles)
    for (int sidx = 0; sidx < t.nslices; sidx++) {
      const uint8_t *ll = P.leadlane.data() + (size_t)(t.slice_base + sidx) * cfs_plan::kLanes;
      const int p0 = sidx * cfs_plan::kLanes;
      auto packets = [&](int l) {
        return p0 + l < (int)t.nvrows ? (int)(P.rowinfo[t.vrow_off + p0 + l] >> 16) : 0;
      };
      int run = 0;
      for (int l = 0; l < cfs_plan::kLanes; l++) {
        const bool give = ll[l] & 64, take = ll[l] & 128;
        if (give) {
          run++;
          if (l == 0 || (l & 15) == 0 || (ll[l] & 63) == l || (ll[l] & 63) != (ll[l - 1] & 63) ||
              packets(l) > packets(l - 1) || run > 2)
            bad++;
        } else {
          run = 0;
        }
        if (take != (l + 1 < cfs_plan::kLanes && (ll[l + 1] & 64))) bad++;
      }
    }
  // (2) fold + send indices cover every strip entry exactly once and point at
  // a strip entry whose column is the destination row
  {
    std::vector<char> seen((size_t)P.nhalo, 0);
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
    int64_t uncovered = 0;
    for (char s : seen)
      if (!s) uncovered++;
    if (uncovered != P.onesided_slots) bad++; // one-sided slots have nothing to fold or send
  }
  // (3) groups partition the tiles; tiles partition the rows
  {
    if (P.group_ptr.front() != 0 || P.group_ptr.back() != (int)P.tiles.size()) bad++;
    if ((int)P.group_ptr.size() != P.ngroups + 1 || P.ngroups % 8) bad++;
    for (size_t g = 1; g < P.group_ptr.size(); g++)
      if (P.group_ptr[g] < P.group_ptr[g - 1]) bad++;
    // every group runs in exactly one launch slot, inside its own XCD's run of slots
    if ((int)P.launch_order.size() != P.ngroups) {
      bad++;
    } else {
      std::vector<char> seen(P.ngroups, 0);
      const int nper = std::max(1, P.ngroups / 8);
      for (int sl = 0; sl < P.ngroups; sl++) {
        const int g = P.launch_order[sl];
        if (g < 0 || g >= P.ngroups || seen[g] || g / nper != sl / nper) {
          bad++;
          continue;
        }
        seen[g] = 1;
      }
    }
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

int cfs_hip_sym_plan_send_info_f64(int n, const int *rowptr, const int *colind,
                                   const double *values, int nranks, int rank,
                                   const int *row_splits, const cfs_hip_options *opt,
                                   int *send_counts, int *rows, int rows_cap,
                                   int *nrows_out) {
  if (!send_counts || !nrows_out || !row_splits || !rowptr || n < 0 || nranks < 1 || rank < 0 ||
      rank >= nranks)
    return set_err(CFS_HIP_ERR_ARG, "null / bad argument");
  SymPlan<double> P;
  if (!cfs_plan::build_plan<double>(n, rowptr, colind, values, nranks, rank, row_splits,
                                    to_opts(opt), P))
    return set_err(plan_error_code(P.error), P.error);
  for (int r = 0; r < nranks; r++) send_counts[r] = P.send_counts[r];
  *nrows_out = (int)P.send_row.size();
  if (rows) {
    if (rows_cap < *nrows_out) return set_err(CFS_HIP_ERR_ARG, "rows_cap too small");
    for (size_t i = 0; i < P.send_row.size(); i++) rows[i] = P.send_row[i];
  }
  return 0;
}

// ---- general CSR ------------------------------------------------------------
int cfs_hip_csr_create_f64(int nrows, int ncols, const int *rowptr, const int *colind,
                           const double *values, cfs_hip_csr_t *out) {
  return csr_create<double>(nrows, ncols, rowptr, colind, values, out);
}
int cfs_hip_csr_create_f32(int nrows, int ncols, const int *rowptr, const int *colind,
                           const float *values, cfs_hip_csr_t *out) {
  return csr_create<float>(nrows, ncols, rowptr, colind, values, out);
}

int cfs_hip_csr_spmv_async(cfs_hip_csr_t h, void *y, const void *x, void *stream) {
  if (!h || !y || !x) return set_err(CFS_HIP_ERR_ARG, "null argument");
  DeviceGuard g(h->device);
  if (!h->form_measured) {
    int rc = csr_choose_form(h, y, x, (hipStream_t)stream);
    if (rc) return rc;
  }
  return csr_launch(h, y, x, (hipStream_t)stream);
}
int cfs_hip_csr_kernel_form(cfs_hip_csr_t h, int *form, int *measured) {
  if (!h || !form) return set_err(CFS_HIP_ERR_ARG, "null argument");
  *form = h->block_form ? CFS_HIP_CSR_FORM_BLOCK : CFS_HIP_CSR_FORM_WAVE;
  if (measured) *measured = h->form_measured ? 1 : 0;
  return 0;
}

int cfs_hip_csr_stats(cfs_hip_csr_t h, int64_t *bytes_streamed, int64_t *narrow_nnz) {
  if (!h) return set_err(CFS_HIP_ERR_ARG, "null argument");
  const int64_t vb = h->value_bytes, nz = h->nnz;
  int64_t b = nz * vb + ((int64_t)h->nrows + 1) * 4;
  const int64_t nar = h->wide == 2 ? h->narrow_nnz : 0;
  if (h->block_form) b += nar * 2 + (nz - nar) * 4 + (int64_t)h->nblocks * (h->cbase.p ? 20 : 4);
  else b += nz * 4 + (int64_t)h->nchunks * 16;
  if (bytes_streamed) *bytes_streamed = b;
  if (narrow_nnz) *narrow_nnz = h->block_form && h->wide == 2 ? h->narrow_nnz : 0;
  return 0;
}

int cfs_hip_csr_spmv(cfs_hip_csr_t h, void *y, const void *x) {
  if (!h || !y || !x) return set_err(CFS_HIP_ERR_ARG, "null argument");
  const size_t vb = (size_t)h->value_bytes;
  return run_staged(h->device, h->stage, (size_t)h->ncols * vb, (size_t)h->nrows * vb, y, x,
                    [&](void *yd, const void *xd, hipStream_t st) {
                      return cfs_hip_csr_spmv_async(h, yd, xd, (void *)st);
                    });
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
