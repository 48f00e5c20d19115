// cfs_devplan.hpp -- tune() on the GPU: the tile schedule of cfs_plan.hpp built by HIP
// kernels from ONE upload of the caller's CSR.
//
// What the reference does at this point, on the host, per thread:
//   lower / diagonal split            include/matrix/csr_matrix.tpp:1218-1348
//   conflict analysis + colouring     :1427-1477, :2009-2363
//   per-thread range lists            :1544-1627          (timed: bench_spmv_mmf.cpp:145-148)
// What cfs_plan.hpp::Builder does on the host for this build: row counts, tile cut,
// virtual rows + slices + slot leaders, packet / COO packing, halo fold index.
// Here the same schedule -- bit for bit, cfs_hip_sym_debug_digest compares the two -- is
// built where it will be used:
//   dp_keys_kernel      every stored lower entry -> (schedule row, column) key + the
//                       position of its value; the diagonal apart; row order checked
//   hipcub radix sort   = the lower triangle in schedule space (clustered order: the
//                       scatter to the later end's row and the per-row sort in one pass;
//                       a mirrored shard's one-sided entries arrive from the rows above)
//   dp_rows_kernel      row pointers, columns, duplicate check
//   dp_cut_kernel       chunks -> LDS-sized tiles: one wave per chunk, greedy under
//                       the slot budget with an LDS hash set of the halo columns
//   dp_vrows_kernel     virtual rows: split long rows, sibling groups (wave scans), stable
//                       sort of the groups (LDS bitonic), slices sorted by packet count
//   dp_leaders_kernel   per slice: prefix leaders (64 x 64 sequence matches, one wave),
//                       sibling chains, stream / slot stream offsets
//   dp_fill_kernel      per tile: halo slot table (LDS hash -> bitonic sort), packets,
//                       COO leftovers: the value MAP and the 16-bit slot stream
//   hipcub sort + dp_fold_* : halo fold index and its records
//   cfs_value_scatter_kernel: the numbers, through the map
// The host keeps what is sequential and small: the clustering sweep (cluster_rows), the
// cost prefix over rows, chunk boundaries, offsets over ~500 tiles, the launch order.
//   Format::hyb         dp_markfar / dp_resolve: far marks per tile (LDS hash with use counts),
//                       the near entries compacted (scan + scatter) for everything above, the
//                       far sections from the flags (lower ends in place, mirrored ends by a sort)
// Not covered (the host builder takes those, the same schedule either way): rows that are
// not sorted by column or hold duplicates, rows longer than 4 096.
#pragma once

#include <hipcub/hipcub.hpp>

#include <thread>

namespace cfs_dev {

using cfs_plan::kLanes;
using cfs_plan::SliceMeta;
using cfs_plan::Tile;
using cfs_rt::DevBuf;

constexpr int kHashSize = 16384;       // LDS hash set of a tile's halo columns (max_slots <= 10 240)
constexpr int kLongRow = 4096;         // rows longer than this take the host builder
constexpr int kBlock = 256;

// what makes the device build hand over to the host builder (each a counter)
enum Flag { F_UNSORTED = 0, F_DUP, F_LONGROW, F_DENSEROW, F_OVERFLOW, F_HALO, F_LRP, F_COUNT };
// global counters
enum Ctr { C_MIRROR = 0, C_CHAINED, C_LANEPK, C_FARCAND, C_COUNT };

__device__ __forceinline__ unsigned hash_col(int c) { return ((unsigned)c * 2654435761u) >> 18; } // 14 bits
// insert column c into the LDS hash set; 1 when it was new
__device__ __forceinline__ int hash_insert(int *hash, int c) {
  const int key = c + 1;
  unsigned h = hash_col(c);
  for (int probe = 0; probe < kHashSize; ++probe) {
    const int old = atomicCAS(&hash[h], 0, key);
    if (old == 0) return 1;
    if (old == key) return 0;
    h = (h + 1) & (kHashSize - 1);
  }
  return 0; // table full: cannot happen below kHashSize entries (long rows are refused)
}

// ---- every stored entry of the block -> (schedule row, column) key, value position --------
// rows [row_lo, row_hi) of the caller's CSR are read (a mirrored shard also reads the rows
// above it: their lower entries (r, c), c in the block, are its one-sided entries (c, r));
// lrp = prefix of the per-row counts of entries with col <= row (the host found them by
// binary search): the position of a value in the COMPACT value array that was uploaded.
__global__ void __launch_bounds__(kBlock)
    dp_keys_kernel(int row_lo, int row_hi, int rb, int re, int mirror, const int32_t *__restrict__ rowptr,
                   const int32_t *__restrict__ colind, const int32_t *__restrict__ lrp,
                   const int32_t *__restrict__ inv, uint64_t *__restrict__ keys, int32_t *__restrict__ vals,
                   int32_t *__restrict__ dsrc, int *__restrict__ flags) {
  const int sub = threadIdx.x & 15;
  const long long ngrp = ((long long)gridDim.x * kBlock) >> 4;
  const uint64_t dummy = (uint64_t)(unsigned)(re - rb) << 32;
  for (long long i = row_lo + (((long long)blockIdx.x * kBlock + threadIdx.x) >> 4); i < row_hi; i += ngrp) {
    const int b = rowptr[i], e = rowptr[i + 1];
    const int lb = lrp[i - row_lo], llen = lrp[i + 1 - row_lo] - lb;
    int nle = 0;
    for (int j = b + sub; j < e; j += 16) {
      const int c = colind[j];
      if (j + 1 < e && colind[j + 1] <= c) atomicAdd(&flags[F_UNSORTED], 1);
      if (c > (int)i) continue;
      nle++;
      const int k = j - b;
      if (k >= llen) continue; // counted below as a mismatch
      const int cp = lb + k;
      uint64_t key = dummy;
      if (c == (int)i) {
        if (i < re) dsrc[(inv ? inv[i - rb] : (int)i) - rb] = cp;
      } else if (i < re) {
        const int p = inv ? inv[i - rb] : (int)i;
        if (c < rb) {
          key = ((uint64_t)(unsigned)(p - rb) << 32) | (unsigned)c;
        } else {
          const int pc = inv ? inv[c - rb] : c;
          const int hi = p > pc ? p : pc, lo = p > pc ? pc : p;
          key = ((uint64_t)(unsigned)(hi - rb) << 32) | (unsigned)lo;
        }
      } else if (mirror && c >= rb && c < re) {
        const int pc = inv ? inv[c - rb] : c;
        key = ((uint64_t)(unsigned)(pc - rb) << 32) | (unsigned)i;
      }
      keys[cp] = key;
      vals[cp] = cp;
    }
    nle += __shfl_xor(nle, 8, 16);
    nle += __shfl_xor(nle, 4, 16);
    nle += __shfl_xor(nle, 2, 16);
    nle += __shfl_xor(nle, 1, 16);
    if (sub == 0 && nle != llen) atomicAdd(&flags[F_LRP], 1);
  }
}

// sorted keys -> row pointers, columns; duplicates; one-sided entries of a mirrored shard
__global__ void __launch_bounds__(kBlock)
    dp_rows_kernel(const uint64_t *__restrict__ keys, long long nl, int rows, int re, int32_t *__restrict__ brp,
                   int32_t *__restrict__ bci, int *__restrict__ flags, unsigned long long *__restrict__ ctr) {
  const long long q = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (q > nl) return;
  int rq = rows;
  uint64_t k = 0;
  if (q < nl) {
    k = keys[q];
    rq = (int)min((unsigned long long)(k >> 32), (unsigned long long)rows);
  }
  int rprev = -1;
  uint64_t kp = 0;
  if (q > 0) {
    kp = keys[q - 1];
    rprev = (int)min((unsigned long long)(kp >> 32), (unsigned long long)rows);
  }
  for (int r = rprev + 1; r <= rq; ++r) brp[r] = (int32_t)q;
  if (q < nl && rq < rows) {
    const int c = (int)(unsigned)(k & 0xffffffffu);
    bci[q] = c;
    if (q > 0 && kp == k) atomicAdd(&flags[F_DUP], 1);
    if (c >= re) atomicAdd(&ctr[C_MIRROR], 1ull);
  }
}

__global__ void __launch_bounds__(kBlock)
    dp_lcnt_kernel(int rows, const int32_t *__restrict__ brp, const int32_t *__restrict__ bci,
                   int32_t *__restrict__ lcnt, int32_t *__restrict__ firstcol, int *__restrict__ flags) {
  const int r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows) return;
  const int b = brp[r], len = brp[r + 1] - b;
  lcnt[r] = len;
  firstcol[r] = len > 0 ? bci[b] : -1;
  if (len > kLongRow) atomicAdd(&flags[F_LONGROW], 1);
}

// ---- chunks -> tiles (cfs_plan::Builder::cut_tiles) ------------------------------------------
// One workgroup per chunk walks its rows in order; the distinct columns outside the tile
// (left of its first row, or right of the block: a mirrored shard) are an LDS hash set.
// Both cuts of the host builder: the greedy one, and -- when it made more than one tile --
// a second one under an even cost cap; the host keeps the one with fewer tiles.
struct CutTile {
  int32_t row0, nown, nslots;
};
// One WAVE per chunk (no workgroup barrier): the row pointers and costs of 64 rows are fetched
// at a time and handed out by shuffles, a row's entries are hashed by the 64 lanes, the new
// columns counted by ballot.
__global__ void __launch_bounds__(64)
    dp_cut_kernel(const int32_t *__restrict__ chunk, int rb, int re, const int32_t *__restrict__ brp,
                  const int32_t *__restrict__ bci, const long long *__restrict__ cost, int max_slots,
                  long long max_tile_nnz, CutTile *__restrict__ out, int *__restrict__ out_count,
                  int *__restrict__ flags, const int32_t *__restrict__ brp_all) {
  // (brp_all: HYB -- brp / bci hold the NEAR entries only, the far ones take no slot; the
  // length of a row that the entry cap and the 65 535 limit see still counts them)
  extern __shared__ int dp_hash[];
  const int lane = threadIdx.x, g = blockIdx.x;
  const int r0 = chunk[g], r1 = chunk[g + 1];
  if (lane == 0) out_count[2 * g] = out_count[2 * g + 1] = 0;
  if (r0 >= r1) return;
  auto cut = [&](long long cap, CutTile *o) -> int {
    int nt = 0, row = r0;
    // row pointers / costs of rows [base, base + 64): lane r holds row base + r (and the end of it)
    int base = r0 - 64, pb = 0, pe = 0, pt = 0;
    long long pc = 0;
    auto row_info = [&](int rw, int &b, int &len, int &tot, long long &c1) {
      if (rw >= base + 64 || rw < base) {
        base = rw;
        const int q = min(rw + lane, r1 - 1) - rb;
        pb = brp[q];
        pe = brp[q + 1];
        pt = brp_all ? brp_all[q + 1] - brp_all[q] : pe - pb;
        pc = cost[q + 1];
      }
      b = __shfl(pb, rw - base);
      len = __shfl(pe, rw - base) - b;
      tot = __shfl(pt, rw - base);
      c1 = __shfl(pc, rw - base);
    };
    while (row < r1) {
      for (int h = lane; h < kHashSize; h += 64) dp_hash[h] = 0;
      const int row0 = row;
      int nown = 0, nhalo = 0;
      long long nnz = 0;
      const long long c0 = cost[row - rb];
      while (row < r1) {
        int b, len, tot;
        long long c1;
        row_info(row, b, len, tot, c1);
        int mynew = 0;
        for (int q = b + lane; q < b + len; q += 64) {
          const int c = bci[q];
          if (c < row0 || c >= re) mynew += hash_insert(dp_hash, c);
        }
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) mynew += __shfl_xor(mynew, o2);
        const int newh = mynew;
        const bool fits = (nown + 1 + nhalo + newh <= max_slots) && (nnz + tot <= max_tile_nnz || nown == 0) &&
                          nown < 65535 && (c1 - c0 <= cap || nown == 0);
        if (tot > 65535 && lane == 0) atomicAdd(&flags[F_DENSEROW], 1);
        if (!fits) {
          if (nown == 0) {
            if (lane == 0) atomicAdd(&flags[F_DENSEROW], 1);
            return -1;
          }
          break;
        }
        nown++;
        nhalo += newh;
        nnz += tot;
        row++;
      }
      if (lane == 0) o[nt] = CutTile{row0, nown, nown + nhalo}; // (at most one tile per row: o has r1 - r0 places)
      nt++;
    }
    return nt;
  };
  // a cut makes at most one tile per row: the chunk's two lists live at 2 x (its first local row)
  CutTile *oa = out + 2 * (size_t)(r0 - rb), *ob = oa + (r1 - r0);
  const int na = cut((long long)1 << 60, oa);
  if (lane == 0) out_count[2 * g] = na;
  if (na > 1) {
    const long long cc = cost[r1 - rb] - cost[r0 - rb];
    const int nb = cut(cc / na + cc / 64 + 1, ob);
    if (lane == 0) out_count[2 * g + 1] = nb;
  }
}

// position of value j of lane l in a packet of cnt lanes (cfs_plan::packet_val_pos)
template <int VS> // sizeof(V)
__device__ __forceinline__ int pkt_val_pos(int l, int j, int cnt) {
  return VS == 8 ? (j >> 1) * 2 * cnt + l * 2 + (j & 1) : l * 4 + j;
}

// ---- HYB (Format::hyb): which entries leave the tile format? ------------------------------------
// cfs_plan::Builder::mark_far for the tiles of a first cut: an entry whose column is an in-block
// halo column of its tile (left of the tile's first row) that the tile uses at most `thr` times.
// One workgroup per tile; LDS: hash set of the halo columns [kHashSize] + a use count each.
__device__ __forceinline__ int hash_slot(int *hash, int c) { // index of c in the set (inserted if new)
  const int key = c + 1;
  unsigned h = hash_col(c);
  for (int probe = 0; probe < kHashSize; ++probe) {
    const int old = atomicCAS(&hash[h], 0, key);
    if (old == 0 || old == key) return (int)h;
    h = (h + 1) & (kHashSize - 1);
  }
  return 0;
}
__global__ void __launch_bounds__(kBlock)
    dp_markfar_kernel(const Tile *__restrict__ tiles, int rb, int thr, const int32_t *__restrict__ brp,
                      const int32_t *__restrict__ bci, uint8_t *__restrict__ far,
                      unsigned long long *__restrict__ ctr) {
  extern __shared__ int dp_lds[];
  int *hash = dp_lds, *uses = dp_lds + kHashSize;
  const Tile t = tiles[blockIdx.x];
  const int tid = threadIdx.x, row0 = t.row0, lr0 = row0 - rb;
  for (int h = tid; h < kHashSize; h += kBlock) hash[h] = uses[h] = 0;
  __syncthreads();
  const int eb = brp[lr0], ee = brp[lr0 + t.nown];
  for (int q = eb + tid; q < ee; q += kBlock) {
    const int c = bci[q];
    if (c < row0 && c >= rb) atomicAdd(&uses[hash_slot(hash, c)], 1);
  }
  __syncthreads();
  unsigned long long mine = 0;
  for (int q = eb + tid; q < ee; q += kBlock) {
    const int c = bci[q];
    if (c < row0 && c >= rb && uses[hash_slot(hash, c)] <= thr) {
      far[q] = 1;
      mine++;
    }
  }
  if (mine) atomicAdd(&ctr[C_FARCAND], mine);
}
__global__ void __launch_bounds__(kBlock)
    dp_tileofrow_kernel(const Tile *__restrict__ tiles, int rb, int32_t *__restrict__ tor) {
  const Tile t = tiles[blockIdx.x];
  for (int r = threadIdx.x; r < t.nown; r += kBlock) tor[t.row0 - rb + r] = blockIdx.x;
}
// cfs_plan::Builder::resolve_far: a marked entry whose two rows ended up in one tile is an ordinary
// entry again; farL / farU = far entries of a row as the lower / the mirrored (upper) end
__global__ void __launch_bounds__(kBlock)
    dp_resolve_kernel(int rows, int rb, const int32_t *__restrict__ brp, const int32_t *__restrict__ bci,
                      const int32_t *__restrict__ tor, uint8_t *__restrict__ far, int32_t *__restrict__ farL,
                      int32_t *__restrict__ farU, unsigned long long *__restrict__ kept) {
  const int r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows) return;
  const int mine = tor[r];
  int low = 0;
  for (int q = brp[r]; q < brp[r + 1]; ++q) {
    if (!far[q]) continue;
    const int c = bci[q] - rb;
    if (tor[c] == mine) {
      far[q] = 0;
      continue;
    }
    low++;
    atomicAdd(&farU[c], 1);
  }
  farL[r] = low;
  if (low) atomicAdd(kept, (unsigned long long)low);
}
// the NEAR entries of the schedule-space matrix, compacted: everything downstream of the cut
// (virtual rows, leaders, packets, slot tables, fold index) reads these arrays
__global__ void __launch_bounds__(kBlock)
    dp_nearflag_kernel(const uint8_t *__restrict__ far, long long nst, int32_t *__restrict__ nf) {
  const long long q = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (q <= nst) nf[q] = q < nst && !far[q] ? 1 : 0;
}
__global__ void __launch_bounds__(kBlock)
    dp_compact_kernel(const uint8_t *__restrict__ far, const int32_t *__restrict__ pos, long long nst,
                      const int32_t *__restrict__ bci, const int32_t *__restrict__ bsrc,
                      int32_t *__restrict__ bci_n, int32_t *__restrict__ bsrc_n) {
  const long long q = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (q >= nst || far[q]) return;
  bci_n[pos[q]] = bci[q];
  bsrc_n[pos[q]] = bsrc[q];
}
__global__ void __launch_bounds__(kBlock)
    dp_nearrows_kernel(int rows, const int32_t *__restrict__ brp, const int32_t *__restrict__ pos,
                       int32_t *__restrict__ brp_n) {
  const int r = blockIdx.x * kBlock + threadIdx.x;
  if (r <= rows) brp_n[r] = pos[brp[r]];
}
// far sections (cfs_plan::Builder::fill_streams): per tile the far entries of its own rows in row /
// entry order [0, nfar_low), then the mirror images of other tiles' far entries that end in its
// rows, sorted by (row, original column) [nfar_low, nfar); 256-entry packets, packet value layout.
// PL / PU = exclusive prefixes of farL / farU over the rows.
template <int VS>
__global__ void __launch_bounds__(kBlock)
    dp_farlow_kernel(int rows, int rb, const Tile *__restrict__ tiles, const int32_t *__restrict__ tor,
                     const int32_t *__restrict__ brp, const int32_t *__restrict__ bci,
                     const int32_t *__restrict__ bsrc, const uint8_t *__restrict__ far,
                     const int32_t *__restrict__ PL, const int32_t *__restrict__ perm,
                     uint16_t *__restrict__ frows, int32_t *__restrict__ fcols, int32_t *__restrict__ fval_map,
                     uint64_t *__restrict__ upkey, int32_t *__restrict__ uppay) {
  const int r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows) return;
  const Tile t = tiles[tor[r]];
  const int lr0 = t.row0 - rb;
  int e = PL[r] - PL[lr0], g = PL[r];
  const unsigned orig_r = (unsigned)(perm ? perm[r] : rb + r);
  for (int q = brp[r]; q < brp[r + 1]; ++q) {
    if (!far[q]) continue;
    const int c = bci[q];
    const int pk = e >> 8, l = (e & 255) >> 2, j = e & 3;
    const long long at = (long long)t.far_off + (long long)pk * 256;
    frows[at + l * 4 + j] = (uint16_t)(r - lr0);
    fcols[at + l * 4 + j] = perm ? perm[c - rb] : c;
    fval_map[at + pkt_val_pos<VS>(l, j, 64)] = bsrc[q];
    upkey[g] = ((uint64_t)(unsigned)(c - rb) << 32) | orig_r;
    uppay[g] = bsrc[q];
    ++e;
    ++g;
  }
}
template <int VS>
__global__ void __launch_bounds__(kBlock)
    dp_farup_kernel(long long K, int rb, const Tile *__restrict__ tiles, const int32_t *__restrict__ tor,
                    const uint64_t *__restrict__ skey, const int32_t *__restrict__ spay,
                    const int32_t *__restrict__ PU, uint16_t *__restrict__ frows, int32_t *__restrict__ fcols,
                    int32_t *__restrict__ fval_map) {
  const long long u = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (u >= K) return;
  const uint64_t k = skey[u];
  const int cl = (int)(k >> 32);
  const Tile t = tiles[tor[cl]];
  const int lr0 = t.row0 - rb;
  const int e = t.nfar_low + (int)(u - PU[lr0]);
  const int pk = e >> 8, l = (e & 255) >> 2, j = e & 3;
  const long long at = (long long)t.far_off + (long long)pk * 256;
  frows[at + l * 4 + j] = (uint16_t)(cl - lr0);
  fcols[at + l * 4 + j] = (int32_t)(unsigned)(k & 0xffffffffu);
  fval_map[at + pkt_val_pos<VS>(l, j, 64)] = spay[u];
}

// ---- per tile: packet cap of a virtual row, virtual rows, COO leftovers ---------------------
__global__ void __launch_bounds__(kBlock)
    dp_tilecount_kernel(const Tile *__restrict__ tiles, int rb, const int32_t *__restrict__ lcnt,
                        int32_t *__restrict__ t_acap, int32_t *__restrict__ t_nvrows,
                        int32_t *__restrict__ t_ncoo) {
  __shared__ long long s_sum;
  __shared__ int s_nvr, s_coo, s_acap;
  const Tile t = tiles[blockIdx.x];
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_sum = 0;
    s_nvr = s_coo = 0;
  }
  __syncthreads();
  long long sum = 0;
  int coo = 0;
  for (int r = tid; r < t.nown; r += kBlock) {
    const int l = lcnt[t.row0 - rb + r];
    sum += l >> 2;
    coo += l & 3;
  }
  atomicAdd((unsigned long long *)&s_sum, (unsigned long long)sum);
  atomicAdd(&s_coo, coo);
  __syncthreads();
  if (tid == 0) {
    const int avg = (int)((s_sum + t.nown - 1) / max(1, (int)t.nown));
    s_acap = max(8, 2 * avg);
  }
  __syncthreads();
  const int acap = s_acap;
  int nvr = 0;
  for (int r = tid; r < t.nown; r += kBlock) {
    const int a = lcnt[t.row0 - rb + r] >> 2;
    nvr += a > acap ? (a + acap - 1) / acap : 1;
  }
  atomicAdd(&s_nvr, nvr);
  __syncthreads();
  if (tid == 0) {
    t_acap[blockIdx.x] = acap;
    t_nvrows[blockIdx.x] = s_nvr;
    t_ncoo[blockIdx.x] = s_coo;
  }
}

// ascending bitonic sort of n (a power of two) unsigned keys in LDS by the whole workgroup
__device__ __forceinline__ void bitonic_sort_lds(unsigned *a, int n, int tid, int nthreads) {
  for (int k = 2; k <= n; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = tid; i < n; i += nthreads) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned x = a[i], y = a[ixj];
          const bool up = (i & k) == 0;
          if ((x > y) == up) {
            a[i] = y;
            a[ixj] = x;
          }
        }
      }
    }
  __syncthreads();
}
__device__ __forceinline__ int pow2_at_least(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}
__device__ __forceinline__ unsigned long long lanes_below(int l) { return l >= 64 ? ~0ull : ((1ull << l) - 1ull); }

// ---- virtual rows of a tile (cfs_plan::Builder::build_vrows), 256 threads ---------------------
// scratch (global, indexed by local row): s_tpos = first position of the row's virtual rows
// in row order, s_gid = its sibling group; per group (at row0 - rb + tile index + group):
// g_tpos (first position, + one sentinel), g_key (longest member), g_dest (position after
// the sort of the groups).  vr_u_* : the virtual rows before the per-slice sort.
__global__ void __launch_bounds__(kBlock)
    dp_vrows_kernel(const Tile *__restrict__ tiles, int rb, const int32_t *__restrict__ lcnt,
                    const int32_t *__restrict__ firstcol, const int32_t *__restrict__ t_acap,
                    int32_t *__restrict__ s_tpos, int32_t *__restrict__ s_gid, int32_t *__restrict__ g_tpos,
                    int32_t *__restrict__ g_key, int32_t *__restrict__ g_dest, uint32_t *__restrict__ vr_u_info,
                    int32_t *__restrict__ vr_u_k0, uint32_t *__restrict__ rowinfo, int32_t *__restrict__ vk0) {
  extern __shared__ unsigned dp_sort[]; // up to 16 384 composite keys
  __shared__ int s_ngroups, s_part[kBlock];
  const int ti = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Tile t = tiles[ti];
  const int acap = t_acap[ti], nown = t.nown, lr0 = t.row0 - rb;
  int32_t *gt = g_tpos + lr0 + ti, *gk = g_key + lr0 + ti, *gd = g_dest + lr0 + ti;
  for (int r = tid; r <= nown; r += kBlock) gk[r] = 0;
  __syncthreads();
  // ---- phase 1 (wave 0): positions and sibling groups, 64 rows per step -----------------------
  if (wave == 0) {
    int carry_tpos = 0, carry_gid = -1, carry_k = 0;
    for (int base = 0; base < nown; base += 64) {
      const int r = base + lane;
      const bool valid = r < nown;
      const int a = valid ? lcnt[lr0 + r] >> 2 : 0;
      const int fc = valid ? firstcol[lr0 + r] : -1;
      const bool split = a > acap;
      const int parts = !valid ? 0 : (split ? (a + acap - 1) / acap : 1);
      const bool hasprev = valid && r > 0;
      const int ap = hasprev ? lcnt[lr0 + r - 1] >> 2 : 0;
      const int fcp = hasprev ? firstcol[lr0 + r - 1] : -1;
      // joins the group of the row before it (same first stored column: siblings)
      const bool b = hasprev && !split && !(ap > acap) && fc >= 0 && fc == fcp;
      const unsigned long long bm = __ballot(b);
      // k = position inside the run of joined rows (0 for the row that starts the run)
      const unsigned long long starts_le = ~bm & (lanes_below(lane) | (1ull << lane));
      int k;
      if (starts_le) k = lane - (63 - __clzll((long long)starts_le));
      else k = carry_k + lane + 1;
      const bool start = valid && (!b || (k & 15) == 0); // a group holds at most 16 rows
      const unsigned long long sm = __ballot(start);
      const int gid = carry_gid + __popcll(sm & (lanes_below(lane) | (1ull << lane)));
      int incl = parts; // inclusive wave scan of parts
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
      }
      const int tpos = carry_tpos + incl - parts;
      if (valid) {
        s_tpos[lr0 + r] = tpos;
        s_gid[lr0 + r] = gid;
        if (start) gt[gid] = tpos;
        atomicMax(&gk[gid], split ? (a + parts - 1) / parts : a); // longest member / chunk
      }
      carry_tpos += __shfl(incl, 63);
      carry_gid += __popcll(sm);
      carry_k = __shfl(k, 63);
    }
    if (lane == 0) {
      gt[carry_gid + 1] = carry_tpos; // sentinel: end of the last group
      s_ngroups = carry_gid + 1;
    }
  }
  __syncthreads();
  const int G = s_ngroups, P2 = pow2_at_least(max(G, 2));
  // ---- phase 2: stable sort of the groups by their longest member, descending -------------------
  for (int i = tid; i < P2; i += kBlock)
    dp_sort[i] = i < G ? (((unsigned)(16383 - min(gk[i], 16383))) << 16) | (unsigned)i : 0xffffffffu;
  bitonic_sort_lds(dp_sort, P2, tid, kBlock);
  // ---- phase 3: destination of every group = exclusive scan of the counts in sorted order --------
  const int per = (G + kBlock - 1) / kBlock;
  {
    int sum = 0;
    for (int i = tid * per; i < min(G, (tid + 1) * per); ++i) {
      const int gi = (int)(dp_sort[i] & 0xffffu);
      sum += gt[gi + 1] - gt[gi];
    }
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int i = 0; i < kBlock; ++i) {
        const int v = s_part[i];
        s_part[i] = run;
        run += v;
      }
    }
    __syncthreads();
    int off = s_part[tid];
    for (int i = tid * per; i < min(G, (tid + 1) * per); ++i) {
      const int gi = (int)(dp_sort[i] & 0xffffu);
      gd[gi] = off;
      off += gt[gi + 1] - gt[gi];
    }
  }
  __syncthreads();
  // ---- phase 4: the virtual rows, in group order ---------------------------------------------------
  for (int r = tid; r < nown; r += kBlock) {
    const int a = lcnt[lr0 + r] >> 2;
    const int gid = s_gid[lr0 + r];
    const int dst = t.vrow_off + gd[gid] + (s_tpos[lr0 + r] - gt[gid]);
    if (a > acap) {
      const int parts = (a + acap - 1) / acap;
      for (int q = 0, k0 = 0; q < parts; ++q) {
        const int ca = (a - k0 + (parts - q) - 1) / (parts - q); // even chunks
        vr_u_info[dst + q] = (uint32_t)r | ((uint32_t)ca << 16);
        vr_u_k0[dst + q] = k0;
        k0 += ca;
      }
    } else {
      vr_u_info[dst] = (uint32_t)r | ((uint32_t)a << 16);
      vr_u_k0[dst] = 0;
    }
  }
  __syncthreads();
  // ---- phase 5: inside a slice, longest first (stable) ---------------------------------------------
  const int nvr = t.nvrows, nsl = (nvr + 63) / 64;
  for (int s = wave; s < nsl; s += kBlock / 64) {
    const int p = s * 64 + lane;
    const bool valid = p < nvr;
    const uint32_t info = valid ? vr_u_info[t.vrow_off + p] : 0u;
    const int k0 = valid ? vr_u_k0[t.vrow_off + p] : 0;
    const int a = valid ? (int)(info >> 16) : -1;
    int rank = 0;
    for (int j = 0; j < 64; ++j) {
      const int aj = __shfl(a, j);
      if (aj > a || (aj == a && j < lane)) rank++;
    }
    if (valid) {
      rowinfo[t.vrow_off + s * 64 + rank] = info;
      vk0[t.vrow_off + s * 64 + rank] = k0;
    }
  }
}

// ---- per slice: slot leaders, sibling chains; per tile: stream offsets -------------------------------
// (cfs_plan::Builder::size_streams, second half)
__global__ void __launch_bounds__(kBlock)
    dp_leaders_kernel(const Tile *__restrict__ tiles, int rb, const int32_t *__restrict__ brp,
                      const int32_t *__restrict__ bci, const uint32_t *__restrict__ rowinfo,
                      const int32_t *__restrict__ vk0, int combine, SliceMeta *__restrict__ slice_meta,
                      uint8_t *__restrict__ leadlane, long long *__restrict__ t_len,
                      long long *__restrict__ t_slen, int32_t *__restrict__ t_rounds,
                      unsigned long long *__restrict__ ctr, int *__restrict__ flags) {
  extern __shared__ int dp_sl[]; // [nslices] value entries, [nslices] slot entries, [nslices] longest lane
  const int ti = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Tile t = tiles[ti];
  const int nvr = t.nvrows, nsl = t.nslices, lr0 = t.row0 - rb;
  int *s_vlen = dp_sl, *s_slen = dp_sl + nsl, *s_amax = dp_sl + 2 * nsl;
  for (int s = wave; s < nsl; s += kBlock / 64) {
    const int p = s * 64 + lane;
    const bool valid = p < nvr;
    const uint32_t info = valid ? rowinfo[t.vrow_off + p] : 0u;
    const int r = (int)(info & 0xffffu), a = valid ? (int)(info >> 16) : 0;
    const int base = valid ? brp[lr0 + r] + vk0[t.vrow_off + p] * 4 : 0;
    const int len4 = a * 4;
    const int fc0 = a >= 1 ? bci[base] : -1 - lane;
    // M: the earlier lanes of the slice whose column sequence starts with mine
    unsigned long long M = 0ull;
    for (int j = 0; j < 63; ++j) {
      const int aj = __shfl(a, j);
      if (aj < 1) continue;
      const int fcj = __shfl(fc0, j), basej = __shfl(base, j);
      if (lane > j && a >= 1 && a <= aj && fc0 == fcj) {
        bool eq = true;
        for (int q = 1; q < len4 && eq; ++q) eq = bci[base + q] == bci[basej + q];
        if (eq) M |= 1ull << j;
      }
    }
    const bool is_leader = !valid || a < 1 || M == 0ull;
    const unsigned long long leaders = __ballot(is_leader);
    int lead = lane;
    if (!is_leader) { // the latest leader among the matches (one exists: matching is transitive)
      const unsigned long long ml = M & leaders;
      lead = ml ? 63 - __clzll((long long)ml) : lane;
    }
    // sibling chains: a follower right behind a lane of its group, inside a row of 16 lanes,
    // hands its products to that lane; runs of at most three lanes (two chained ones)
    bool chained = false;
    if (combine) {
      const int leadp = __shfl_up(lead, 1);
      const bool q = valid && lane > 0 && (lane & 15) != 0 && lead != lane && lead == leadp;
      const unsigned long long qm = __ballot(q);
      if (q) {
        const unsigned long long z = ~qm & lanes_below(lane); // lane 0 is never chained: z != 0
        const int lz = 63 - __clzll((long long)z);
        chained = ((lane - lz - 1) % 3) != 2;
      }
    }
    const unsigned long long cf = __ballot(chained);
    const bool take = lane + 1 < 64 && ((cf >> (lane + 1)) & 1ull);
    leadlane[(size_t)(t.slice_base + s) * 64 + lane] = (uint8_t)((lead & 63) | (chained ? 64 : 0) | (take ? 128 : 0));
    // sizes
    int v = valid ? len4 : 0, sl = (valid && is_leader) ? len4 : 0, ch = chained ? a : 0, al = valid ? a : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      v += __shfl_xor(v, o);
      sl += __shfl_xor(sl, o);
      ch += __shfl_xor(ch, o);
      al += __shfl_xor(al, o);
    }
    const int cnt0 = __popcll(__ballot(a >= 1));
    if (lane == 0) {
      s_vlen[s] = v;
      s_slen[s] = sl;
      s_amax[s] = a; // lanes are sorted: lane 0 is the longest
      SliceMeta &sm = slice_meta[t.slice_base + s];
      sm.leaders = leaders;
      sm.soff_cnt0 = (uint32_t)cnt0 << 25; // offset added below
      if (combine) {
        atomicAdd(&ctr[C_CHAINED], (unsigned long long)ch);
        atomicAdd(&ctr[C_LANEPK], (unsigned long long)al);
      }
    }
  }
  __syncthreads();
  if (tid == 0) { // offsets of the slices inside the tile's streams (a few hundred slices)
    long long off = 0, soff = 0;
    int rounds = 0;
    for (int s = 0; s < nsl; ++s) {
      off = (off + 7) / 8 * 8;
      soff = (soff + 3) / 4 * 4;
      SliceMeta &sm = slice_meta[t.slice_base + s];
      sm.voff = (uint32_t)off;
      if (soff >= (1 << 25)) atomicAdd(&flags[F_OVERFLOW], 1);
      sm.soff_cnt0 |= (uint32_t)soff & 0x1ffffffu;
      off += s_vlen[s];
      soff += s_slen[s];
      rounds += s_amax[s];
    }
    t_len[ti] = (off + 7) / 8 * 8;
    t_slen[ti] = (soff + 7) / 8 * 8;
    t_rounds[ti] = rounds;
  }
}

// ---- per tile: halo slot table, packets, COO leftovers (cfs_plan::Builder::fill_streams) -----
// Values are not touched here: every stored value of the device format gets the POSITION of
// its number in the uploaded value array (val_map / cval_map / diag_map); the numbers follow
// through cfs_value_scatter_kernel -- the same kernel that refreshes them later.
template <int VS>
__global__ void __launch_bounds__(kBlock)
    dp_fill_kernel(Tile *__restrict__ tiles, int rb, int re, int mirror, int far_thr,
                   const int32_t *__restrict__ brp, const int32_t *__restrict__ bci,
                   const int32_t *__restrict__ bsrc, const int32_t *__restrict__ dsrc,
                   const int32_t *__restrict__ lcnt, const uint32_t *__restrict__ rowinfo,
                   const int32_t *__restrict__ vk0, const SliceMeta *__restrict__ slice_meta,
                   const uint8_t *__restrict__ leadlane, int32_t *__restrict__ halo_col,
                   int32_t *__restrict__ val_map, uint16_t *__restrict__ slots, int32_t *__restrict__ cval_map,
                   uint16_t *__restrict__ crows, uint16_t *__restrict__ ccols, int32_t *__restrict__ diag_map,
                   unsigned long long *__restrict__ ctr, int *__restrict__ flags) {
  extern __shared__ int dp_lds[]; // hash [kHashSize], then the halo list [kHashSize]
  int *hash = dp_lds;
  unsigned *list = reinterpret_cast<unsigned *>(dp_lds + kHashSize);
  __shared__ int s_n, s_part[kBlock];
  const int ti = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Tile t = tiles[ti];
  const int nown = t.nown, row0 = t.row0, lr0 = row0 - rb;
  for (int h = tid; h < kHashSize; h += kBlock) hash[h] = 0;
  if (tid == 0) s_n = 0;
  __syncthreads();
  const int eb = brp[lr0], ee = brp[lr0 + nown]; // the tile's entries are contiguous
  for (int q = eb + tid; q < ee; q += kBlock) {
    const int c = bci[q];
    if (c < row0 || c >= re) hash_insert(hash, c);
  }
  __syncthreads();
  // halo list: in-block columns first (y window + strip entry), then the off-block columns
  // of a mirrored shard (x only), each class ascending
  for (int h = tid; h < kHashSize; h += kBlock)
    if (hash[h]) {
      const int c = hash[h] - 1;
      const bool offb = mirror && (c < rb || c >= re);
      list[atomicAdd(&s_n, 1)] = (offb ? 0x80000000u : 0u) | (unsigned)c;
    }
  __syncthreads();
  const int nh = s_n;
  if (nh != t.nslots - nown) {
    if (tid == 0) atomicAdd(&flags[F_HALO], 1);
    return;
  }
  const int P2 = pow2_at_least(max(nh, 2));
  for (int i = nh + tid; i < P2; i += kBlock) list[i] = 0xffffffffu;
  bitonic_sort_lds(list, P2, tid, kBlock);
  auto halo_index = [&](unsigned key) { // position of a halo column in the sorted list
    int lo = 0, hi = nh;
    while (lo < hi) {
      const int m = (lo + hi) >> 1;
      if (list[m] < key) lo = m + 1;
      else hi = m;
    }
    return lo;
  };
  auto slot_of = [&](int c) -> uint16_t {
    if (c >= row0 && c < re) return (uint16_t)(c - row0);
    const bool offb = mirror && (c < rb || c >= re);
    return (uint16_t)(nown + halo_index((offb ? 0x80000000u : 0u) | (unsigned)c));
  };
  for (int h = tid; h < nh; h += kBlock) halo_col[t.halo_off + h] = (int32_t)(list[h] & 0x7fffffffu);
  if (tid == 0) tiles[ti].ny = nown + halo_index(0x80000000u);
  // what would HYB take out?  in-block halo columns this tile uses at most far_thr times
  if (far_thr > 0) {
    __syncthreads();
    for (int h = tid; h < nh; h += kBlock) hash[h] = 0; // reuse: uses per halo column
    __syncthreads();
    for (int q = eb + tid; q < ee; q += kBlock) {
      const int c = bci[q];
      if (c < row0 && c >= rb) atomicAdd(&hash[halo_index((unsigned)c)], 1);
    }
    __syncthreads();
    unsigned long long mine = 0;
    for (int h = tid; h < nh; h += kBlock)
      if (hash[h] > 0 && hash[h] <= far_thr) mine += (unsigned long long)hash[h];
    if (mine) atomicAdd(&ctr[C_FARCAND], mine);
  }
  // packets: one wave per slice
  const int nvr = t.nvrows, nsl = t.nslices;
  for (int s = wave; s < nsl; s += kBlock / 64) {
    const int p = s * 64 + lane;
    const bool valid = p < nvr;
    const uint32_t info = valid ? rowinfo[t.vrow_off + p] : 0u;
    const int r = (int)(info & 0xffffu), a = valid ? (int)(info >> 16) : 0;
    const int k0 = valid ? vk0[t.vrow_off + p] : 0;
    const int base = valid ? brp[lr0 + r] + k0 * 4 : 0;
    if (valid) diag_map[t.vrow_off + p] = k0 == 0 ? dsrc[lr0 + r] : -1;
    const SliceMeta sm = slice_meta[t.slice_base + s];
    const bool is_leader = (sm.leaders >> lane) & 1ull;
    const int rank = __popcll(sm.leaders & lanes_below(lane));
    const int amax = __shfl(a, 0);
    long long o = t.nnz_off + (long long)sm.voff, os = t.sl_off + (long long)(sm.soff_cnt0 & 0x1ffffffu);
    for (int g = 0; g < amax; ++g) {
      const int cnt = __popcll(__ballot(a > g));
      const int nlead = __popcll(sm.leaders & lanes_below(cnt));
      if (a > g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = base + g * 4 + j;
          val_map[o + pkt_val_pos<VS>(lane, j, cnt)] = bsrc[q];
          if (is_leader) slots[os + rank * 4 + j] = slot_of(bci[q]);
        }
      }
      o += 4 * (long long)cnt;
      os += 4 * (long long)nlead;
    }
  }
  // COO leftovers: the last len % 4 entries of every row, row order, 256-entry packets
  const int per = (nown + kBlock - 1) / kBlock;
  {
    int sum = 0;
    for (int r = tid * per; r < min(nown, (tid + 1) * per); ++r) sum += lcnt[lr0 + r] & 3;
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int i = 0; i < kBlock; ++i) {
        const int v = s_part[i];
        s_part[i] = run;
        run += v;
      }
    }
    __syncthreads();
    int e = s_part[tid];
    for (int r = tid * per; r < min(nown, (tid + 1) * per); ++r) {
      const int len = lcnt[lr0 + r], b = brp[lr0 + r];
      for (int k = (len >> 2) << 2; k < len; ++k, ++e) {
        const int pk = e >> 8, l = (e & 255) >> 2, j = e & 3;
        const long long at = (long long)t.coo_off + (long long)pk * 256;
        cval_map[at + pkt_val_pos<VS>(l, j, 64)] = bsrc[b + k];
        crows[at + l * 4 + j] = (uint16_t)r;
        ccols[at + l * 4 + j] = slot_of(bci[b + k]);
      }
    }
  }
}

// slot -> the caller's column (own rows and halo), cfs_plan::Builder::finish
__global__ void __launch_bounds__(kBlock)
    dp_slotcol_kernel(const Tile *__restrict__ tiles, int rb, int re, const int32_t *__restrict__ perm,
                      const int32_t *__restrict__ halo_col, int32_t *__restrict__ slot_col) {
  const Tile t = tiles[blockIdx.x];
  for (int i = threadIdx.x; i < t.nslots; i += kBlock) {
    const int c = i < t.nown ? t.row0 + i : halo_col[t.halo_off + (i - t.nown)];
    slot_col[t.slot_off + i] = (perm && c >= rb && c < re) ? perm[c - rb] : c;
  }
}

// deterministic build: scale exponent of every slot = that of its (original) matrix row
__global__ void __launch_bounds__(kBlock)
    dp_slotexp_kernel(const int32_t *__restrict__ slot_col, const int16_t *__restrict__ row_exp, long long nsl,
                      int16_t *__restrict__ slot_exp) {
  const long long q = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (q < nsl) slot_exp[q] = row_exp[slot_col[q]];
}
__global__ void __launch_bounds__(kBlock)
    dp_gatherkeys_kernel(const uint32_t *__restrict__ skeys, const int32_t *__restrict__ start, int m,
                         int32_t *__restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < m) out[i] = (int32_t)skeys[start[i]];
}

// ---- halo fold index: strips -> destination rows (cfs_plan::Builder::fold_index + the records
//      of make_fold_records) ------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
    dp_foldkeys_kernel(const int32_t *__restrict__ halo_col, int H, uint32_t *__restrict__ keys,
                       int32_t *__restrict__ vals) {
  const int q = blockIdx.x * kBlock + threadIdx.x;
  if (q >= H) return;
  keys[q] = (uint32_t)halo_col[q];
  vals[q] = q;
}
// sorted columns: first of a run of equal columns inside [lo, hi) = one fold destination
__global__ void __launch_bounds__(kBlock)
    dp_foldflag_kernel(const uint32_t *__restrict__ skeys, int lo, int hi, int32_t *__restrict__ flag) {
  const int i = lo + blockIdx.x * kBlock + threadIdx.x;
  if (i >= hi) return;
  flag[i - lo] = (i == lo || skeys[i] != skeys[i - 1]) ? 1 : 0;
}
__global__ void __launch_bounds__(kBlock)
    dp_foldstart_kernel(const int32_t *__restrict__ flag, const int32_t *__restrict__ pos, int F,
                        int32_t *__restrict__ start) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i > F) return;
  if (i == F) start[pos[F - 1] + flag[F - 1]] = F; // sentinel behind the last destination
  else if (flag[i]) start[pos[i]] = i;
}
__global__ void __launch_bounds__(kBlock)
    dp_foldlen_kernel(const int32_t *__restrict__ start, int nfold, int32_t *__restrict__ restlen) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= nfold) return;
  const int len = start[i + 1] - start[i];
  restlen[i] = len > 3 ? len - 1 : 0;
}
__global__ void __launch_bounds__(kBlock)
    dp_foldrec_kernel(const uint32_t *__restrict__ skeys, const int32_t *__restrict__ sidx, int lo,
                      const int32_t *__restrict__ start, const int32_t *__restrict__ restoff, int nfold, int rb,
                      int re, const int32_t *__restrict__ perm, int4 *__restrict__ rec,
                      int32_t *__restrict__ rest) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i > nfold) return;
  if (i == nfold) {
    rec[i] = make_int4(0, 0, -1, -1); // padding record
    return;
  }
  const int b = start[i], len = start[i + 1] - b;
  const int c = (int)skeys[lo + b];
  const int dst = ((perm && c >= rb && c < re) ? perm[c - rb] : c) - rb;
  const int32_t *idx = sidx + lo + b;
  int4 r = make_int4(dst, idx[0], len > 1 ? idx[1] : -1, len == 3 ? idx[2] : -1);
  if (len > 3) {
    const int ro = restoff[i];
    r.w = -(ro + 2);
    rest[ro] = len - 2;
    for (int k = 2; k < len; ++k) rest[ro + k - 1] = idx[k];
  }
  rec[i] = r;
}

// value positions in the COMPACT array -> positions in the caller's values[] (keep_value_map)
__global__ void __launch_bounds__(kBlock)
    dp_map_to_caller_kernel(int32_t *__restrict__ map, long long n, const int32_t *__restrict__ lrp, int nrows_up,
                            int row_lo, const int32_t *__restrict__ rowptr) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const int cp = map[i];
    if (cp < 0) continue;
    int lo = 0, hi = nrows_up; // last row with lrp[row] <= cp
    while (lo < hi) {
      const int m = (lo + hi + 1) >> 1;
      if (lrp[m] <= cp) lo = m;
      else hi = m - 1;
    }
    // (empty rows share a prefix value: the LAST of them up to the owner is found, which is the
    // row that really holds an entry only if it is non-empty; step over empty rows)
    while (lo + 1 <= nrows_up && lrp[lo + 1] <= cp) ++lo;
    map[i] = rowptr[row_lo + lo] + (cp - lrp[lo]);
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct Scratch { // hipcub temporary storage, grown on demand
  DevBuf buf;
  int need(size_t bytes) {
    if (buf.bytes >= bytes && buf.p) return 0;
    return buf.alloc(bytes + (bytes >> 2) + 256);
  }
};

// 1 = use the host builder (the reason is in `why`), 0 = built, < 0 = error (set_err)
constexpr int kUseHost = 1;

// one row order of the block on the device: its lower triangle in schedule space
struct Sched {
  int rb = 0, re = 0, rows = 0;
  long long nl = 0, nst = 0; // compact entries read / entries stored in the schedule
  DevBuf inv, perm;          // orig - rb -> schedule row; schedule row - rb -> orig (empty: identity)
  DevBuf keys, keys2, kv, kv2;
  DevBuf brp, bci, dsrc, lcnt, firstcol;
  const int32_t *bsrc() const { return (const int32_t *)kv2.p; }
  // the big arrays of a placement (first-time device allocations of this size take tens of
  // milliseconds: build() reserves them on a thread of its own while the host clusters)
  int reserve(long long nl_, int rows_) {
    const size_t nl = (size_t)nl_;
    int rc;
    if ((rc = keys.alloc((nl + 1) * 8)) || (rc = keys2.alloc((nl + 1) * 8)) || (rc = kv.alloc((nl + 1) * 4)) ||
        (rc = kv2.alloc((nl + 1) * 4)) || (rc = dsrc.alloc((size_t)rows_ * 4 + 4)) ||
        (rc = brp.alloc(((size_t)rows_ + 2) * 4)) || (rc = bci.alloc((nl + 1) * 4)) ||
        (rc = lcnt.alloc((size_t)rows_ * 4 + 4)) || (rc = firstcol.alloc((size_t)rows_ * 4 + 4)))
      return rc;
    return 0;
  }
  std::vector<int32_t> h_lcnt, chunk;
  std::vector<int64_t> cost;
  std::vector<Tile> tiles;
  std::vector<int32_t> group_ptr;
  long long nhalo = 0, mirror_entries = 0;
};

// the uploaded caller matrix: full structure, compact lower + diagonal values
template <typename V> struct Input {
  int n = 0, row_lo = 0, row_hi = 0; // rows read: [row_lo, row_hi)
  DevBuf rowptr, colind, lrp, lva;
  std::vector<int32_t> h_lrp;        // [row_hi - row_lo + 1]
  long long nl = 0, nnz_low = 0, nnz_diag = 0;
};

// what a second build of the same rows in the same (clustered) order reuses: tune() tries two
// window shapes, the clusters of the coarser schedule are pairs of the finer one's -- same row
// order, same schedule-space matrix, no second upload
template <typename V> struct Kept {
  Input<V> in; // the upload: every later build of these rows reads it (tune() makes up to three)
  bool mirror = false;
  // the last clustered placement and the clusters it belongs to (a build with the same clusters, or
  // with pairs of them, reuses it: no sweep of the host, no sort)
  bool has_sc = false;
  int nc = 0, max_slots = 0;
  bool cost_model = false;
  std::vector<int32_t> perm, chunk;
  Sched SC;
};

// positions of the entries with col <= row (binary search: columns ascend -- verified on the
// device against the real counts), compact value array, uploads
template <typename V>
void scan_input(int n, const int *rowptr, const int *colind, int rb, int re, bool mirror, Input<V> &in) {
  in.n = n;
  in.row_lo = rb;
  in.row_hi = mirror ? n : re;
  const int nr = in.row_hi - in.row_lo;
  in.h_lrp.assign((size_t)nr + 1, 0);
  long long low = 0, dg = 0;
#pragma omp parallel for schedule(static) reduction(+ : low, dg) num_threads(cfs_plan::host_threads())
  for (int i = in.row_lo; i < in.row_hi; i++) {
    const int *b = colind + rowptr[i], *e = colind + rowptr[i + 1];
    const int *d = std::upper_bound(b, e, i); // first column > i
    in.h_lrp[(size_t)(i - in.row_lo) + 1] = (int32_t)(d - b);
    if (i < re) {
      const bool has = d > b && d[-1] == i;
      low += (d - b) - (has ? 1 : 0);
      dg += has ? 1 : 0;
    }
  }
  for (int r = 0; r < nr; r++) in.h_lrp[r + 1] += in.h_lrp[r];
  in.nl = in.h_lrp[nr];
  in.nnz_low = low;
  in.nnz_diag = dg;
}
template <typename V>
int upload_input(int n, const int *rowptr, const int *colind, const V *values, Input<V> &in) {
  int rc;
  if ((rc = in.rowptr.upload(rowptr, ((size_t)n + 1) * 4))) return rc;
  if ((rc = in.colind.upload(colind, (size_t)rowptr[n] * 4))) return rc;
  if ((rc = in.lrp.upload(in.h_lrp.data(), in.h_lrp.size() * 4))) return rc;
  // compact values: the row prefixes, copied by all host threads into one array
  {
    cfs_plan::BigVec<V> lva;
    lva.resize((size_t)in.nl + 1);
#pragma omp parallel for schedule(static) num_threads(cfs_plan::host_threads())
    for (int i = in.row_lo; i < in.row_hi; i++) {
      const size_t o = (size_t)in.h_lrp[i - in.row_lo], cnt = (size_t)in.h_lrp[i - in.row_lo + 1] - o;
      if (cnt) memcpy(lva.data() + o, values + rowptr[i], cnt * sizeof(V));
    }
    lva[(size_t)in.nl] = V(0);
    if ((rc = in.lva.upload(lva.data(), ((size_t)in.nl + 1) * sizeof(V)))) return rc;
    cfs_plan::release_async(lva);
  }
  return 0;
}

inline int read_flags(const DevBuf &flags, int *out) {
  HIPCHK(hipMemcpy(out, flags.p, F_COUNT * sizeof(int), hipMemcpyDeviceToHost));
  return 0;
}
inline const char *flag_name(int f) {
  static const char *const names[F_COUNT] = {"rows not sorted by column", "duplicate entries", "row longer than 4096",
                                             "dense row / row too long", "offset overflow",
                                             "halo count mismatch", "lower-count mismatch"};
  return names[f];
}
inline bool any_flag(const int *f, std::string &why) {
  for (int k = 0; k < F_COUNT; k++)
    if (f[k]) {
      why = flag_name(k);
      return true;
    }
  return false;
}

// the lower triangle of the block in schedule space, for the row order `perm_h` (NULL: natural)
template <typename V>
int place(const Input<V> &in, int rb, int re, bool mirror, const std::vector<int32_t> *perm_h, Sched &S,
          Scratch &tmp, DevBuf &flags, unsigned long long *h_ctr, DevBuf &ctr, std::string &why) {
  S.rb = rb;
  S.re = re;
  S.rows = re - rb;
  S.nl = in.nl;
  const int rows = S.rows;
  int rc;
  cfs_plan::PhaseTimer pt;
  if (perm_h) {
    std::vector<int32_t> inv((size_t)rows);
#pragma omp parallel for schedule(static) num_threads(cfs_plan::host_threads())
    for (int p = 0; p < rows; p++) inv[(*perm_h)[p] - rb] = rb + p;
    if ((rc = S.inv.upload(inv.data(), inv.size() * 4))) return rc;
    if ((rc = S.perm.upload(perm_h->data(), perm_h->size() * 4))) return rc;
  }
  const size_t nl = (size_t)in.nl;
  if (!S.keys.p || S.keys.bytes < (nl + 1) * 8 || !S.firstcol.p) // (not reserved ahead)
    if ((rc = S.reserve(in.nl, rows))) return rc;
  pt.lap("  place: inverse order, buffers");
  HIPCHK(hipMemsetAsync(S.dsrc.p, 0xff, (size_t)rows * 4 + 4, 0));
  HIPCHK(hipMemsetAsync(ctr.p, 0, C_COUNT * 8, 0));
  const int nr = in.row_hi - in.row_lo;
  if (nr > 0) {
    const int grid = (int)std::min<long long>(((long long)nr * 16 + kBlock - 1) / kBlock, 256 * 64);
    hipLaunchKernelGGL(dp_keys_kernel, dim3(grid), dim3(kBlock), 0, 0, in.row_lo, in.row_hi, rb, re, mirror ? 1 : 0,
                       (const int32_t *)in.rowptr.p, (const int32_t *)in.colind.p, (const int32_t *)in.lrp.p,
                       perm_h ? (const int32_t *)S.inv.p : nullptr, (uint64_t *)S.keys.p, (int32_t *)S.kv.p,
                       (int32_t *)S.dsrc.p, (int *)flags.p);
  }
  int end_bit = 32;
  while (end_bit < 64 && ((unsigned long long)rows >> (end_bit - 32)) != 0) end_bit++;
  if (nl > 0) {
    size_t tb = 0;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const uint64_t *)S.keys.p, (uint64_t *)S.keys2.p,
                                              (const int32_t *)S.kv.p, (int32_t *)S.kv2.p, (int)nl, 0, end_bit,
                                              (hipStream_t)0));
    if ((rc = tmp.need(tb))) return rc;
    tb = tmp.buf.bytes;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(tmp.buf.p, tb, (const uint64_t *)S.keys.p, (uint64_t *)S.keys2.p,
                                              (const int32_t *)S.kv.p, (int32_t *)S.kv2.p, (int)nl, 0, end_bit,
                                              (hipStream_t)0));
  }
  hipLaunchKernelGGL(dp_rows_kernel, dim3((unsigned)((nl + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0,
                     (const uint64_t *)S.keys2.p, (long long)nl, rows, re, (int32_t *)S.brp.p, (int32_t *)S.bci.p,
                     (int *)flags.p, (unsigned long long *)ctr.p);
  if (rows > 0)
    hipLaunchKernelGGL(dp_lcnt_kernel, dim3((rows + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, rows,
                       (const int32_t *)S.brp.p, (const int32_t *)S.bci.p, (int32_t *)S.lcnt.p,
                       (int32_t *)S.firstcol.p, (int *)flags.p);
  HIPCHK(hipGetLastError());
  S.h_lcnt.assign((size_t)rows + 1, 0);
  HIPCHK(hipMemcpy(S.h_lcnt.data(), S.lcnt.p, (size_t)rows * 4, hipMemcpyDeviceToHost));
  pt.lap("  place: keys, sort, rows (device)");
  // the sort's inputs are not needed any more
  S.keys = DevBuf();
  S.kv = DevBuf();
  int f[F_COUNT];
  if ((rc = read_flags(flags, f))) return rc;
  if (any_flag(f, why)) return kUseHost;
  HIPCHK(hipMemcpy(h_ctr, ctr.p, C_COUNT * 8, hipMemcpyDeviceToHost));
  S.mirror_entries = (long long)h_ctr[C_MIRROR];
  int32_t last = 0;
  HIPCHK(hipMemcpy(&last, (const int32_t *)S.brp.p + rows, 4, hipMemcpyDeviceToHost));
  S.nst = last;
  S.keys2 = DevBuf();
  // cost prefix of the rows (cfs_plan::Builder::count_rows, no far entries): in build()
  pt.lap("  place: frees");
  return 0;
}

// chunks -> tiles on the device; fills S.tiles / S.group_ptr / S.nhalo
template <typename V>
int cut_tiles(Sched &S, const cfs_plan::ChunkLayout &L, const cfs_plan::Options &opt, DevBuf &flags,
              std::string &why, const Sched *near = nullptr) {
  // (near: HYB -- the matrix without its far entries, which take no slot)
  const int nc = L.nchunks(), rows = S.rows, rb = S.rb;
  int rc;
  DevBuf d_chunk, d_cost, d_out, d_cnt;
  if ((rc = d_chunk.upload(S.chunk.data(), S.chunk.size() * 4))) return rc;
  if ((rc = d_cost.upload(S.cost.data(), S.cost.size() * 8))) return rc;
  if ((rc = d_out.alloc((size_t)2 * std::max(rows, 1) * sizeof(CutTile)))) return rc;
  if ((rc = d_cnt.alloc((size_t)nc * 2 * 4))) return rc;
  const long long max_tile_nnz = opt.max_tile_nnz > 0 ? opt.max_tile_nnz : (long long)1 << 30;
  HIPCHK(hipFuncSetAttribute((const void *)dp_cut_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                             kHashSize * 4));
  hipLaunchKernelGGL(dp_cut_kernel, dim3(nc), dim3(64), kHashSize * 4, 0, (const int32_t *)d_chunk.p, rb, S.re,
                     (const int32_t *)(near ? near->brp.p : S.brp.p), (const int32_t *)(near ? near->bci.p : S.bci.p),
                     (const long long *)d_cost.p, L.max_slots, max_tile_nnz, (CutTile *)d_out.p, (int *)d_cnt.p,
                     (int *)flags.p, near ? (const int32_t *)S.brp.p : (const int32_t *)nullptr);
  HIPCHK(hipGetLastError());
  std::vector<CutTile> out((size_t)2 * std::max(rows, 1));
  std::vector<int> cnt((size_t)nc * 2);
  HIPCHK(hipMemcpy(out.data(), d_out.p, out.size() * sizeof(CutTile), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cnt.data(), d_cnt.p, cnt.size() * 4, hipMemcpyDeviceToHost));
  int f[F_COUNT];
  if ((rc = read_flags(flags, f))) return rc;
  if (any_flag(f, why)) return kUseHost;
  S.tiles.clear();
  S.group_ptr.assign((size_t)nc + 1, 0);
  S.nhalo = 0;
  for (int g = 0; g < nc; g++) {
    S.group_ptr[g] = (int32_t)S.tiles.size();
    const int na = cnt[2 * g], nb = cnt[2 * g + 1];
    const bool use_b = na > 1 && nb > 0 && nb <= na; // the even cut, unless it needs more tiles
    const int c0 = S.chunk[g] - rb, clen = S.chunk[g + 1] - S.chunk[g];
    const CutTile *src = out.data() + 2 * (size_t)c0 + (use_b ? clen : 0);
    for (int k = 0; k < (use_b ? nb : na); k++) {
      Tile t{};
      t.row0 = src[k].row0;
      t.nown = src[k].nown;
      t.nslots = src[k].nslots;
      S.nhalo += t.nslots - t.nown;
      S.tiles.push_back(t);
    }
  }
  S.group_ptr[nc] = (int32_t)S.tiles.size();
  return 0;
}

template <typename T> inline int dl(std::vector<T> &dst, const DevBuf &src, size_t count) {
  dst.resize(count);
  if (count) HIPCHK(hipMemcpy(dst.data(), src.p, count * sizeof(T), hipMemcpyDeviceToHost));
  return 0;
}

// cost prefix of the rows (cfs_plan::Builder::count_rows); farL / farU: far entries of a row as the
// lower / the mirrored end (NULL: none)
template <typename V> void row_costs(Sched &S, const int32_t *farL, const int32_t *farU) {
  const int rows = S.rows;
  S.cost.assign((size_t)rows + 1, 0);
  for (int r = 0; r < rows; r++) {
    const int64_t fl = farL ? farL[r] : 0, fu = farU ? farU[r] : 0;
    S.cost[r + 1] = S.cost[r] + ((int64_t)S.h_lcnt[r] - fl) * (int64_t)(sizeof(V) + 2) +
                    (farL ? (fl + fu) * (int64_t)(2 * sizeof(V) + 6) : 0) + (int64_t)(4 + 3 * sizeof(V)) +
                    2 * (int64_t)sizeof(V);
  }
}

// HYB state of one row order: which stored entries are far, and the matrix without them
struct FarState {
  bool active = false;         // far entries exist; `near` holds everything else
  long long marked = 0, kept = 0;
  DevBuf far, farL, farU, tor; // flag per stored entry; per row: far entries as lower / upper end; tile of a row
  std::vector<int32_t> h_farL, h_farU;
  Sched near; // brp / bci / kv2 (value positions) / lcnt / firstcol of the near entries
};

// near <- the entries of S that are not flagged
inline int compact_near(const Sched &S, FarState &F, Scratch &tmp, DevBuf &flags) {
  const long long nst = S.nst;
  const int rows = S.rows;
  int rc;
  DevBuf nf, pos;
  if ((rc = nf.alloc(((size_t)nst + 1) * 4)) || (rc = pos.alloc(((size_t)nst + 1) * 4))) return rc;
  hipLaunchKernelGGL(dp_nearflag_kernel, dim3((unsigned)((nst + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0,
                     (const uint8_t *)F.far.p, nst, (int32_t *)nf.p);
  size_t tb = 0;
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (const int32_t *)nf.p, (int32_t *)pos.p, (int)(nst + 1),
                                          (hipStream_t)0));
  if ((rc = tmp.need(tb))) return rc;
  tb = tmp.buf.bytes;
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(tmp.buf.p, tb, (const int32_t *)nf.p, (int32_t *)pos.p, (int)(nst + 1),
                                          (hipStream_t)0));
  int32_t nn = 0;
  HIPCHK(hipMemcpy(&nn, (const int32_t *)pos.p + nst, 4, hipMemcpyDeviceToHost));
  Sched &N = F.near;
  N.rb = S.rb;
  N.re = S.re;
  N.rows = rows;
  N.nl = S.nl;
  N.nst = nn;
  if ((rc = N.brp.alloc(((size_t)rows + 2) * 4)) || (rc = N.bci.alloc(((size_t)nn + 1) * 4)) ||
      (rc = N.kv2.alloc(((size_t)nn + 1) * 4)) || (rc = N.lcnt.alloc((size_t)rows * 4 + 4)) ||
      (rc = N.firstcol.alloc((size_t)rows * 4 + 4)))
    return rc;
  if (nst > 0)
    hipLaunchKernelGGL(dp_compact_kernel, dim3((unsigned)((nst + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0,
                       (const uint8_t *)F.far.p, (const int32_t *)pos.p, nst, (const int32_t *)S.bci.p, S.bsrc(),
                       (int32_t *)N.bci.p, (int32_t *)N.kv2.p);
  hipLaunchKernelGGL(dp_nearrows_kernel, dim3((rows + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, rows,
                     (const int32_t *)S.brp.p, (const int32_t *)pos.p, (int32_t *)N.brp.p);
  if (rows > 0)
    hipLaunchKernelGGL(dp_lcnt_kernel, dim3((rows + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, rows,
                       (const int32_t *)N.brp.p, (const int32_t *)N.bci.p, (int32_t *)N.lcnt.p,
                       (int32_t *)N.firstcol.p, (int *)flags.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize()); // nf / pos go out of scope
  return 0;
}

// tile of every row; far marks of pairs inside one tile cleared; farL / farU / kept
inline int resolve_far(const Sched &S, FarState &F) {
  const int rows = S.rows, T = (int)S.tiles.size();
  int rc;
  DevBuf d_t, d_kept;
  if ((rc = d_t.upload(S.tiles.data(), (size_t)T * sizeof(Tile))) || (rc = d_kept.alloc(8))) return rc;
  if (!F.tor.p && (rc = F.tor.alloc((size_t)rows * 4 + 4))) return rc;
  if (!F.farL.p && ((rc = F.farL.alloc(((size_t)rows + 1) * 4)) || (rc = F.farU.alloc(((size_t)rows + 1) * 4)))) return rc;
  HIPCHK(hipMemsetAsync(F.farL.p, 0, ((size_t)rows + 1) * 4, 0));
  HIPCHK(hipMemsetAsync(F.farU.p, 0, ((size_t)rows + 1) * 4, 0));
  HIPCHK(hipMemsetAsync(d_kept.p, 0, 8, 0));
  hipLaunchKernelGGL(dp_tileofrow_kernel, dim3(T), dim3(kBlock), 0, 0, (const Tile *)d_t.p, S.rb, (int32_t *)F.tor.p);
  hipLaunchKernelGGL(dp_resolve_kernel, dim3((rows + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, rows, S.rb,
                     (const int32_t *)S.brp.p, (const int32_t *)S.bci.p, (const int32_t *)F.tor.p, (uint8_t *)F.far.p,
                     (int32_t *)F.farL.p, (int32_t *)F.farU.p, (unsigned long long *)d_kept.p);
  HIPCHK(hipGetLastError());
  unsigned long long k = 0;
  HIPCHK(hipMemcpy(&k, d_kept.p, 8, hipMemcpyDeviceToHost));
  F.kept = (long long)k;
  if ((rc = dl(F.h_farL, F.farL, (size_t)rows)) || (rc = dl(F.h_farU, F.farU, (size_t)rows))) return rc;
  return 0;
}

// the cut of one row order with Format::hyb -- the sequence of cfs_plan::Builder::run: first cut,
// far marks, cut again without them, resolve; with far entries left: costs (and, in natural
// order, chunk boundaries) once more, final cut, resolve, costs.  S.chunk and the base costs are set.
template <typename V, class NaturalChunks>
int hyb_cut(Sched &S, FarState &F, const cfs_plan::ChunkLayout &L, const cfs_plan::Options &opt, bool natural,
            NaturalChunks &&natural_chunks, Scratch &tmp, DevBuf &flags, DevBuf &ctr, std::string &why) {
  int rc;
  F.active = false;
  F.marked = F.kept = 0;
  if ((rc = cut_tiles<V>(S, L, opt, flags, why))) return rc;
  if (S.rows == 0 || S.nst == 0) return 0;
  const int T1 = (int)S.tiles.size();
  if ((rc = F.far.alloc((size_t)S.nst + 1))) return rc;
  HIPCHK(hipMemsetAsync(F.far.p, 0, (size_t)S.nst + 1, 0));
  HIPCHK(hipMemsetAsync((char *)ctr.p + C_FARCAND * 8, 0, 8, 0));
  {
    DevBuf d_t;
    if ((rc = d_t.upload(S.tiles.data(), (size_t)T1 * sizeof(Tile)))) return rc;
    HIPCHK(hipFuncSetAttribute((const void *)dp_markfar_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               2 * kHashSize * 4));
    hipLaunchKernelGGL(dp_markfar_kernel, dim3(T1), dim3(kBlock), 2 * kHashSize * 4, 0, (const Tile *)d_t.p, S.rb,
                       std::max(1, opt.far_uses), (const int32_t *)S.brp.p, (const int32_t *)S.bci.p,
                       (uint8_t *)F.far.p, (unsigned long long *)ctr.p);
    HIPCHK(hipGetLastError());
    unsigned long long mk = 0;
    HIPCHK(hipMemcpy(&mk, (const char *)ctr.p + C_FARCAND * 8, 8, hipMemcpyDeviceToHost));
    F.marked = (long long)mk;
  }
  if (F.marked == 0) return 0; // (the host cuts once more: the same tiles)
  if ((rc = compact_near(S, F, tmp, flags))) return rc;
  if ((rc = cut_tiles<V>(S, L, opt, flags, why, &F.near))) return rc;
  if ((rc = resolve_far(S, F))) return rc;
  if (F.kept == 0) return 0;
  row_costs<V>(S, F.h_farL.data(), F.h_farU.data());
  if (natural) natural_chunks(S);
  if ((rc = compact_near(S, F, tmp, flags))) return rc;
  if ((rc = cut_tiles<V>(S, L, opt, flags, why, &F.near))) return rc;
  if ((rc = resolve_far(S, F))) return rc;
  if (F.kept == 0) {
    row_costs<V>(S, nullptr, nullptr);
    return 0;
  }
  row_costs<V>(S, F.h_farL.data(), F.h_farU.data());
  if ((rc = compact_near(S, F, tmp, flags))) return rc;
  F.active = true;
  return 0;
}

__global__ void dp_bounds_kernel(const uint32_t *__restrict__ skeys, int H, unsigned rb, unsigned re,
                                 int32_t *__restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  auto lb = [&](unsigned key) {
    int lo = 0, hi = H;
    while (lo < hi) {
      const int m = (lo + hi) >> 1;
      if (skeys[m] < key) lo = m + 1;
      else hi = m;
    }
    return lo;
  };
  out[0] = lb(rb);
  out[1] = lb(re);
}


// what the caller (sym_create) may reuse for a second build with half as many groups:
// the clustered row order (cfs_plan::ScheduleSpace holds perm / chunk; device_only marks that
// its matrix arrays were never built on the host)

// Build the schedule of rows [rb, re) on the device and hand its arrays to `m` (a
// SymMatrix<V>).  Returns 0 (built), kUseHost (why is set: the host builder takes it), or a
// negative error code.
template <typename V, class SymMatrixT>
int build(int n, const int *rowptr, const int *colind, const V *values, int nranks, int rank,
          const int *row_splits_in, const cfs_plan::Options &opt, SymMatrixT &m,
          cfs_plan::ScheduleSpace<V> *cache, std::string &why) {
  using namespace cfs_plan;
  PhaseTimer pt;
  const int rb = row_splits_in ? row_splits_in[rank] : 0;
  const int re = row_splits_in ? row_splits_in[rank + 1] : n;
  if (rb < 0 || re > n || rb > re) return kUseHost; // the host builder reports it
  const int rows = re - rb;
  const bool mirror = opt.mirror_offblock && nranks > 1;
  if (!opt.group_share.empty() || rows < 1 ||
      (opt.block_threads != 0 && opt.block_threads != 256 && opt.block_threads != 512 && opt.block_threads != 1024)) {
    why = "option not covered by the device builder";
    return kUseHost;
  }
  if (mirror) { // same refusal as build_plan: the images of the rows above must pair up
    int64_t up = 0, low_in = 0;
#pragma omp parallel for schedule(static) reduction(+ : up) num_threads(host_threads())
    for (int i = rb; i < re; i++) {
      const int *b = colind + rowptr[i], *e = colind + rowptr[i + 1];
      up += e - std::lower_bound(b, e, re);
    }
#pragma omp parallel for schedule(static) reduction(+ : low_in) num_threads(host_threads())
    for (int r = re; r < n; r++) {
      const int *b = colind + rowptr[r], *e = colind + rowptr[r + 1];
      low_in += std::lower_bound(b, e, re) - std::lower_bound(b, e, rb);
    }
    if (up != low_in) {
      why = "mirror: structurally unsymmetric";
      return kUseHost;
    }
  }
  const ChunkLayout L = chunk_layout<V>(rows, opt);
  const int nc = L.nchunks();
  int rc;
  DevBuf flags, ctr;
  if ((rc = flags.alloc(F_COUNT * sizeof(int))) || (rc = ctr.alloc(C_COUNT * 8))) return rc;
  HIPCHK(hipMemset(flags.p, 0, F_COUNT * sizeof(int)));
  unsigned long long h_ctr[C_COUNT] = {0};
  Scratch tmp;
  const bool may_cluster = opt.reorder && opt.force_order != 1 && rows >= 256;
  // what an earlier build of these rows left on the device (tune() builds up to three schedules:
  // two window shapes, Format::hyb): the upload always; the clustered placement when this build
  // has the same clusters (same count, window and cost model) or pairs of them (the second shape
  // after a clustered first one -- the rule of cfs_plan::build_plan)
  Kept<V> *KP = nullptr;
  if (cache && cache->device_keep && cache->rb == rb && cache->re == re) {
    KP = static_cast<Kept<V> *>(cache->device_keep.get());
    if (KP->mirror != mirror || !KP->in.rowptr.p || !KP->in.colind.p) KP = nullptr;
  }
  const bool same_clusters = KP && may_cluster && KP->has_sc && KP->nc == nc && KP->max_slots == L.max_slots &&
                             KP->cost_model == opt.cost_model;
  const bool pair_clusters = KP && may_cluster && KP->has_sc && cache->valid && cache->nchunks == 2 * nc &&
                             KP->nc == 2 * nc && !same_clusters;
  Kept<V> *K = (same_clusters || pair_clusters) ? KP : nullptr; // (the placement too)
  Input<V> in_local;
  Sched SC_local, SN, *S = nullptr;
  Input<V> &in = KP ? KP->in : in_local;
  Sched &SC = K ? K->SC : SC_local;
  auto natural_chunks = [&](Sched &X) { // cfs_plan::Builder::cut_chunks
    X.chunk.assign((size_t)nc + 1, re);
    X.chunk[0] = rb;
    const std::vector<double> share = L.shares(opt);
    std::vector<double> cum((size_t)nc + 1, 0.0);
    for (int c = 0; c < nc; c++) cum[c + 1] = cum[c] + share[c];
    for (int c = 1; c < nc; c++) {
      const int64_t target = (int64_t)((double)X.cost[rows] * (cum[c] / cum[nc]));
      int r = (int)(std::lower_bound(X.cost.begin(), X.cost.end(), target) - X.cost.begin());
      if (r > rows) r = rows;
      if (rb + r < X.chunk[c - 1]) r = X.chunk[c - 1] - rb;
      X.chunk[c] = rb + r;
    }
    X.chunk[nc] = re;
  };
  FarState FC, FN; // Format::hyb: the far entries of either order
  auto cut_with = [&](Sched &X, FarState &F, bool natural, Scratch &tmp_, DevBuf &flags_, DevBuf &ctr_,
                      std::string &why_) -> int {
    row_costs<V>(X, nullptr, nullptr); // (a kept placement may hold the costs of an earlier HYB build)
    if (natural) natural_chunks(X);
    if (!opt.hyb) return cut_tiles<V>(X, L, opt, flags_, why_);
    return hyb_cut<V>(X, F, L, opt, natural, natural_chunks, tmp_, flags_, ctr_, why_);
  };
  auto cut = [&](Sched &X, FarState &F, bool natural) -> int { return cut_with(X, F, natural, tmp, flags, ctr, why); };

  // The natural order needs no clusters: when the host is about to sweep for them (tens to hundreds of
  // milliseconds, nothing for the device to do), the uploader thread goes on to place and cut the natural
  // order -- with buffers of its own -- as soon as the matrix is on the device.
  const bool will_sweep = may_cluster && !same_clusters &&
                          !(cache && cache->valid && cache->rb == rb && cache->re == re && cache->nchunks == 2 * nc &&
                            (pair_clusters || !cache->device_only));
  const bool nat_early = !KP && will_sweep && opt.force_order != 2 && rows > 0 && !getenv("CFS_HIP_NO_EARLY_NATURAL");
  bool nat_done = false;
  int nat_place_rc = 0, nat_rc = 0; // place()'s result; the cut's
  std::string nat_why, nat_err;
  DevBuf flagsN, ctrN;
  Scratch tmpN;
  unsigned long long h_ctrN[C_COUNT] = {0};
  // the upload of the caller's CSR (PCIe, all host threads copy into the page-locked pieces)
  // runs beside the clustering sweep of the host, which only reads the caller's arrays
  int up_rc = 0, cur_dev = 0;
  std::string up_err;
  HIPCHK(hipGetDevice(&cur_dev));
  if (!KP) scan_input<V>(n, rowptr, colind, rb, re, mirror, in); // row prefixes col <= row (binary searches)
  auto do_upload = [&]() {
    (void)hipSetDevice(cur_dev);
    up_rc = upload_input<V>(n, rowptr, colind, values, in);
    if (up_rc) {
      up_err = cfs_rt::last_error(); // (the message is thread-local)
      return;
    }
    if (!nat_early) return;
    int r = flagsN.alloc(F_COUNT * sizeof(int));
    if (!r) r = ctrN.alloc(C_COUNT * 8);
    if (!r && hipMemset(flagsN.p, 0, F_COUNT * sizeof(int)) != hipSuccess) r = CFS_HIP_ERR_DEVICE;
    if (!r) r = place<V>(in, rb, re, mirror, nullptr, SN, tmpN, flagsN, h_ctrN, ctrN, nat_why);
    nat_place_rc = r;
    if (!r) nat_rc = cut_with(SN, FN, true, tmpN, flagsN, ctrN, nat_why);
    if (nat_place_rc < 0 || nat_rc < 0) nat_err = cfs_rt::last_error();
    nat_done = true;
  };
  std::thread uploader;
  bool threaded = false;
  if (!KP) {
    try {
      uploader = std::thread(do_upload);
      threaded = true;
    } catch (...) {
      do_upload();
    }
  }
  struct Joiner { // every return path waits for the helper threads
    std::thread &t;
    ~Joiner() {
      if (t.joinable()) t.join();
    }
  } joiner{uploader};
  // ... and so does the first-time allocation of the placement's arrays (~50 ms for 2.3 GB)
  int res_rc = 0;
  std::thread reserver;
  if (!K) {
    try {
      reserver = std::thread([&]() {
        (void)hipSetDevice(cur_dev);
        res_rc = (may_cluster ? SC : SN).reserve(in.nl, rows); // the order that is placed first
      });
    } catch (...) {
    }
  }
  Joiner joiner2{reserver};

  // ---- row order: natural, or the clusters of the host's graph-growing sweep -------------------
  std::vector<int32_t> perm, cchunk;
  bool have_clusters = false;
  if (may_cluster) {
    if (same_clusters) { // the sweep would find the same clusters again
      perm = KP->perm;
      cchunk = KP->chunk;
      have_clusters = true;
    } else if (cache && cache->valid && cache->rb == rb && cache->re == re && cache->nchunks == 2 * nc &&
               (pair_clusters || !cache->device_only)) {
      perm = cache->perm; // the coarser schedule's clusters are pairs of the finer one's
      cchunk.resize((size_t)nc + 1);
      for (int g = 0; g <= nc; g++) cchunk[g] = cache->chunk[2 * g];
      have_clusters = true;
    } else {
      cluster_rows<V>(n, rowptr, colind, rb, re, nc, L.shares(opt), perm, cchunk, mirror, L.cluster_cost(opt));
      have_clusters = true;
      if (cache) {
        std::shared_ptr<void> keepalive = cache->device_keep; // (the upload outlives the clusters)
        cache->drop();
        cache->device_keep = keepalive;
        cache->perm = perm;
        cache->chunk = cchunk;
        cache->rb = rb;
        cache->re = re;
        cache->nchunks = nc;
        cache->valid = true;
        cache->device_only = true;
      }
    }
  }
  if (threaded) uploader.join();
  if (reserver.joinable()) reserver.join();
  if (up_rc) return set_err(up_rc, up_err);
  if (res_rc) return res_rc;
  pt.lap("device: upload CSR || cluster_rows (host)");
  bool use_clustered = have_clusters;
  const bool reused = have_clusters && !same_clusters && cache && cache->valid && cache->nchunks == 2 * nc;
  if (have_clusters) {
    if (!K) {
      rc = place<V>(in, rb, re, mirror, &perm, SC, tmp, flags, h_ctr, ctr, why);
      if (rc) return rc;
    } else {
      HIPCHK(hipMemsetAsync(ctr.p, 0, C_COUNT * 8, 0)); // (place() would have)
    }
    SC.chunk = cchunk;
    if ((rc = cut(SC, FC, false)) < 0) return rc;
    const bool c_ok = rc == 0;
    if (!c_ok) HIPCHK(hipMemset(flags.p, 0, F_COUNT * sizeof(int))); // natural order may still do
    pt.lap("device: clustered order placed + cut");
    if (opt.force_order != 2 && !reused) {
      if (nat_done) { // placed and cut beside the clustering sweep
        if (nat_place_rc < 0 || nat_rc < 0) return set_err(nat_place_rc < 0 ? nat_place_rc : nat_rc, nat_err);
        if (nat_place_rc) { // place() itself handed over (unsorted rows, duplicates ...)
          why = nat_why;
          return nat_place_rc;
        }
        rc = nat_rc;
        if (rc) why = nat_why;
      } else {
        rc = place<V>(in, rb, re, mirror, nullptr, SN, tmp, flags, h_ctr, ctr, why);
        if (rc) return rc;
        if ((rc = cut(SN, FN, true)) < 0) return rc;
      }
      const bool n_ok = rc == 0;
      if (!n_ok && !c_ok) return kUseHost;
      if (!n_ok) HIPCHK(hipMemset(flags.p, 0, F_COUNT * sizeof(int)));
      if (getenv("CFS_PLAN_VERBOSE"))
        fprintf(stderr, "[cfs_dev] halo slots: clustered %lld (%zu tiles), natural %lld (%zu tiles)\n",
                c_ok ? SC.nhalo : -1LL, SC.tiles.size(), n_ok ? SN.nhalo : -1LL, SN.tiles.size());
      // the rules of cfs_plan::build_plan
      if (!c_ok) use_clustered = false;
      else if (n_ok && (sizeof(V) == 8 ? 2 * SN.nhalo <= 5 * SC.nhalo : SN.nhalo <= 6 * SC.nhalo))
        use_clustered = false;
      else if (n_ok && SN.tiles.size() <= SC.tiles.size())
        use_clustered = false;
      pt.lap("device: natural order placed + cut");
    } else if (!c_ok) {
      return kUseHost;
    }
    if (cache && cache->device_only) cache->valid = cache->valid && use_clustered; // reusable only if kept
  }
  const std::vector<int32_t> perm_all = (cache && have_clusters && !K) ? perm : std::vector<int32_t>(); // for Kept
  if (use_clustered) {
    S = &SC;
    SN = Sched();
  } else if (K && !same_clusters) {
    return kUseHost; // (cannot happen: pairs of kept clusters are only offered for the clustered order)
  } else {
    if (!SN.rows && rows) {
      rc = place<V>(in, rb, re, mirror, nullptr, SN, tmp, flags, h_ctr, ctr, why);
      if (rc) return rc;
      if ((rc = cut(SN, FN, true))) return rc;
      pt.lap("device: natural order placed + cut");
    }
    S = &SN;
    if (!cache) SC_local = Sched(); // (a cache keeps the clustered placement for a later build)
    perm.clear();
  }
  if (!cache) in.colind = DevBuf(); // the structure of the caller's matrix has been read (a cache keeps the upload)
  // Format::hyb: everything below reads the matrix WITHOUT its far entries
  FarState &FS = use_clustered ? FC : FN;
  (use_clustered ? FN : FC) = FarState();
  const Sched *M = FS.active ? &FS.near : S;

  // ---- per tile: virtual rows -------------------------------------------------------------------
  std::vector<Tile> &tiles = S->tiles;
  const int T = (int)tiles.size();
  SymPlan<V> &P = m.P;
  P = SymPlan<V>();
  DevBuf d_tiles, t_acap, t_nvrows, t_ncoo;
  if ((rc = d_tiles.upload(tiles.data(), (size_t)T * sizeof(Tile))) || (rc = t_acap.alloc((size_t)T * 4 + 4)) ||
      (rc = t_nvrows.alloc((size_t)T * 4 + 4)) || (rc = t_ncoo.alloc((size_t)T * 4 + 4)))
    return rc;
  hipLaunchKernelGGL(dp_tilecount_kernel, dim3(T), dim3(kBlock), 0, 0, (const Tile *)d_tiles.p, rb,
                     (const int32_t *)M->lcnt.p, (int32_t *)t_acap.p, (int32_t *)t_nvrows.p, (int32_t *)t_ncoo.p);
  HIPCHK(hipGetLastError());
  std::vector<int32_t> h_nvrows, h_ncoo;
  if ((rc = dl(h_nvrows, t_nvrows, T)) || (rc = dl(h_ncoo, t_ncoo, T))) return rc;
  int64_t halo = 0, slices = 0, nsl = 0, nvr = 0, coo = 0, farlen = 0;
  int max_nsl = 1;
  for (int ti = 0; ti < T; ti++) {
    Tile &t = tiles[ti];
    t.nvrows = h_nvrows[ti];
    t.nslices = (t.nvrows + kLanes - 1) / kLanes;
    t.ncoo = h_ncoo[ti];
    t.nfar = t.nfar_low = 0;
    if (FS.active)
      for (int r = t.row0 - rb; r < t.row0 - rb + t.nown; r++) {
        t.nfar_low += FS.h_farL[r];
        t.nfar += FS.h_farL[r] + FS.h_farU[r];
      }
    t.halo_off = (int32_t)halo;
    t.slice_base = (int32_t)slices;
    t.slot_off = (int32_t)nsl;
    t.vrow_off = (int32_t)nvr;
    t.far_off = (int32_t)farlen;
    farlen += align_up(t.nfar, 256);
    t.coo_off = (int32_t)coo;
    t.aexp = -1000;
    halo += t.nslots - t.nown;
    nsl += t.nslots;
    slices += t.nslices;
    nvr += t.nvrows;
    coo += align_up(t.ncoo, 256);
    max_nsl = std::max(max_nsl, (int)t.nslices);
    P.coo_entries += t.ncoo;
    if (halo > 0x7fffffffLL || nsl > 0x7ffffff0LL || nvr > 0x7ffffff0LL || coo > 0x7fffff00LL || farlen > 0x7fffff00LL) {
      why = "index overflow";
      return kUseHost;
    }
  }
  HIPCHK(hipMemcpy(d_tiles.p, tiles.data(), (size_t)T * sizeof(Tile), hipMemcpyHostToDevice));
  DevBuf s_tpos, s_gid, g_tpos, g_key, g_dest, vr_u_info, vr_u_k0, vk0;
  const size_t gsz = (size_t)rows + T + 2;
  if ((rc = s_tpos.alloc((size_t)rows * 4 + 4)) || (rc = s_gid.alloc((size_t)rows * 4 + 4)) ||
      (rc = g_tpos.alloc(gsz * 4)) || (rc = g_key.alloc(gsz * 4)) || (rc = g_dest.alloc(gsz * 4)) ||
      (rc = vr_u_info.alloc((size_t)nvr * 4 + 4)) || (rc = vr_u_k0.alloc((size_t)nvr * 4 + 4)) ||
      (rc = vk0.alloc((size_t)nvr * 4 + 4)) || (rc = m.rowinfo.alloc(((size_t)nvr + 1) * 4)) ||
      (rc = m.slice_meta.alloc((size_t)slices * sizeof(SliceMeta) + 16)) ||
      (rc = m.leadlane.alloc((size_t)slices * kLanes + kLanes)))
    return rc;
  HIPCHK(hipMemsetAsync(m.rowinfo.p, 0, ((size_t)nvr + 1) * 4, 0));
  HIPCHK(hipMemsetAsync(m.slice_meta.p, 0, (size_t)slices * sizeof(SliceMeta) + 16, 0));
  HIPCHK(hipMemsetAsync(m.leadlane.p, 0, (size_t)slices * kLanes + kLanes, 0));
  HIPCHK(hipFuncSetAttribute((const void *)dp_vrows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4));
  hipLaunchKernelGGL(dp_vrows_kernel, dim3(T), dim3(kBlock), 16384 * 4, 0, (const Tile *)d_tiles.p, rb,
                     (const int32_t *)M->lcnt.p, (const int32_t *)M->firstcol.p, (const int32_t *)t_acap.p,
                     (int32_t *)s_tpos.p, (int32_t *)s_gid.p, (int32_t *)g_tpos.p, (int32_t *)g_key.p,
                     (int32_t *)g_dest.p, (uint32_t *)vr_u_info.p, (int32_t *)vr_u_k0.p, (uint32_t *)m.rowinfo.p,
                     (int32_t *)vk0.p);
  DevBuf t_len, t_slen, t_rounds;
  if ((rc = t_len.alloc((size_t)T * 8 + 8)) || (rc = t_slen.alloc((size_t)T * 8 + 8)) ||
      (rc = t_rounds.alloc((size_t)T * 4 + 4)))
    return rc;
  hipLaunchKernelGGL(dp_leaders_kernel, dim3(T), dim3(kBlock), (size_t)3 * max_nsl * 4, 0, (const Tile *)d_tiles.p, rb,
                     (const int32_t *)M->brp.p, (const int32_t *)M->bci.p, (const uint32_t *)m.rowinfo.p,
                     (const int32_t *)vk0.p, opt.combine_siblings ? 1 : 0, (SliceMeta *)m.slice_meta.p,
                     (uint8_t *)m.leadlane.p, (long long *)t_len.p, (long long *)t_slen.p, (int32_t *)t_rounds.p,
                     (unsigned long long *)ctr.p, (int *)flags.p);
  HIPCHK(hipGetLastError());
  std::vector<long long> h_len, h_slen;
  if ((rc = dl(h_len, t_len, T)) || (rc = dl(h_slen, t_slen, T)) || (rc = dl(P.tile_rounds, t_rounds, T))) return rc;
  {
    int f[F_COUNT];
    if ((rc = read_flags(flags, f))) return rc;
    if (any_flag(f, why)) return kUseHost;
  }
  int64_t off = 0, soff = 0;
  for (int ti = 0; ti < T; ti++) {
    tiles[ti].nnz_off = off;
    off += h_len[ti];
    tiles[ti].sl_off = soff;
    soff += h_slen[ti];
  }
  HIPCHK(hipMemcpy(d_tiles.p, tiles.data(), (size_t)T * sizeof(Tile), hipMemcpyHostToDevice));
  pt.lap("device: virtual rows, leaders, offsets");

  // ---- per tile: slot tables, packets, leftovers ------------------------------------------------
  DevBuf halo_col, val_map, cval_map, diag_map;
  const size_t vlen = (size_t)off + kStreamPad, slen = (size_t)soff + kStreamPad, clen = (size_t)coo + 256;
  if ((rc = halo_col.alloc(((size_t)halo + 1) * 4)) || (rc = val_map.alloc(vlen * 4)) ||
      (rc = cval_map.alloc(clen * 4)) || (rc = diag_map.alloc(((size_t)nvr + 1) * 4)) ||
      (rc = m.crows.alloc(clen * 2)) || (rc = m.ccols.alloc(clen * 2)))
    return rc;
  if ((rc = m.slots.alloc(slen * 2))) return rc;
  HIPCHK(hipMemsetAsync(halo_col.p, 0, ((size_t)halo + 1) * 4, 0));
  HIPCHK(hipMemsetAsync(val_map.p, 0xff, vlen * 4, 0));
  HIPCHK(hipMemsetAsync(cval_map.p, 0xff, clen * 4, 0));
  HIPCHK(hipMemsetAsync(diag_map.p, 0xff, ((size_t)nvr + 1) * 4, 0));
  HIPCHK(hipMemsetAsync(m.slots.p, 0, slen * 2, 0));
  HIPCHK(hipMemsetAsync(m.crows.p, 0, clen * 2, 0));
  HIPCHK(hipMemsetAsync(m.ccols.p, 0, clen * 2, 0));
  {
    const void *fk = sizeof(V) == 8 ? (const void *)dp_fill_kernel<8> : (const void *)dp_fill_kernel<4>;
    HIPCHK(hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kHashSize * 4));
    const int far_thr = opt.count_far ? std::max(1, opt.far_uses) : 0;
    Tile *dt = (Tile *)d_tiles.p;
    const int32_t *a_brp = (const int32_t *)M->brp.p, *a_bci = (const int32_t *)M->bci.p, *a_bsrc = M->bsrc(),
                  *a_dsrc = (const int32_t *)S->dsrc.p, *a_lcnt = (const int32_t *)M->lcnt.p,
                  *a_vk0 = (const int32_t *)vk0.p;
    const uint32_t *a_ri = (const uint32_t *)m.rowinfo.p;
    const SliceMeta *a_sm = (const SliceMeta *)m.slice_meta.p;
    const uint8_t *a_ll = (const uint8_t *)m.leadlane.p;
    int32_t *a_hc = (int32_t *)halo_col.p, *a_vm = (int32_t *)val_map.p, *a_cm = (int32_t *)cval_map.p,
            *a_dm = (int32_t *)diag_map.p;
    uint16_t *a_sl = (uint16_t *)m.slots.p, *a_cr = (uint16_t *)m.crows.p, *a_cc = (uint16_t *)m.ccols.p;
    unsigned long long *a_ctr = (unsigned long long *)ctr.p;
    int *a_fl = (int *)flags.p;
    int a_rb = rb, a_re = re, a_mirror = mirror ? 1 : 0, a_thr = far_thr;
    void *args[] = {&dt,   &a_rb, &a_re, &a_mirror, &a_thr, &a_brp, &a_bci, &a_bsrc, &a_dsrc, &a_lcnt, &a_ri, &a_vk0,
                    &a_sm, &a_ll, &a_hc, &a_vm,     &a_sl,  &a_cm,  &a_cr,  &a_cc,   &a_dm,   &a_ctr,  &a_fl};
    HIPCHK(hipLaunchKernel(fk, dim3(T), dim3(kBlock), args, 2 * kHashSize * 4, 0));
  }
  if ((rc = m.slot_col.alloc(((size_t)nsl + 1) * 4))) return rc;
  HIPCHK(hipMemsetAsync(m.slot_col.p, 0, ((size_t)nsl + 1) * 4, 0));
  hipLaunchKernelGGL(dp_slotcol_kernel, dim3(T), dim3(kBlock), 0, 0, (const Tile *)d_tiles.p, rb, re,
                     perm.empty() ? nullptr : (const int32_t *)S->perm.p, (const int32_t *)halo_col.p,
                     (int32_t *)m.slot_col.p);
  if (opt.deterministic) { // per-slot scale exponents of the fixed-point sums (cfs_plan::compute_row_exp)
    std::vector<int16_t> row_exp;
    compute_row_exp<V>(n, rowptr, values, row_exp);
    DevBuf d_rexp;
    if ((rc = d_rexp.upload(row_exp.data(), row_exp.size() * 2)) || (rc = m.slot_exp.alloc(((size_t)nsl + 1) * 2)))
      return rc;
    HIPCHK(hipMemsetAsync(m.slot_exp.p, 0, ((size_t)nsl + 1) * 2, 0));
    hipLaunchKernelGGL(dp_slotexp_kernel, dim3((unsigned)((nsl + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0,
                       (const int32_t *)m.slot_col.p, (const int16_t *)d_rexp.p, (long long)nsl, (int16_t *)m.slot_exp.p);
    HIPCHK(hipDeviceSynchronize()); // d_rexp goes out of scope
    m.dev_slot_exp = (const short *)m.slot_exp.p;
  }
  HIPCHK(hipGetLastError());
  pt.lap("device: slot tables, packets, leftovers");

  // ---- halo fold index ----------------------------------------------------------------------------
  int nfold = 0;
  long long onesided = 0;
  const int H = (int)halo;
  if (H > 0) {
    DevBuf fk, fk2, fv, fv2, bounds;
    if ((rc = fk.alloc((size_t)H * 4)) || (rc = fk2.alloc((size_t)H * 4)) || (rc = fv.alloc((size_t)H * 4)) ||
        (rc = fv2.alloc((size_t)H * 4)) || (rc = bounds.alloc(8)))
      return rc;
    hipLaunchKernelGGL(dp_foldkeys_kernel, dim3((H + kBlock - 1) / kBlock), dim3(kBlock), 0, 0,
                       (const int32_t *)halo_col.p, H, (uint32_t *)fk.p, (int32_t *)fv.p);
    size_t tb = 0;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const uint32_t *)fk.p, (uint32_t *)fk2.p,
                                              (const int32_t *)fv.p, (int32_t *)fv2.p, H, 0, 32, (hipStream_t)0));
    if ((rc = tmp.need(tb))) return rc;
    tb = tmp.buf.bytes;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(tmp.buf.p, tb, (const uint32_t *)fk.p, (uint32_t *)fk2.p,
                                              (const int32_t *)fv.p, (int32_t *)fv2.p, H, 0, 32, (hipStream_t)0));
    hipLaunchKernelGGL(dp_bounds_kernel, dim3(1), dim3(64), 0, 0, (const uint32_t *)fk2.p, H, (unsigned)rb, (unsigned)re,
                       (int32_t *)bounds.p);
    int32_t hb[2] = {0, 0};
    HIPCHK(hipMemcpy(hb, bounds.p, 8, hipMemcpyDeviceToHost));
    const int lo = hb[0], F = hb[1] - hb[0];
    onesided = mirror ? (long long)H - F : 0;
    if (!mirror && hb[1] != H) {
      why = "halo column right of the block";
      return kUseHost;
    }
    if (!mirror && lo > 0) { // exchange form of a shard: sums for rows of lower ranks are packed and sent
      DevBuf sflag, spos, sstart, srow;
      if ((rc = sflag.alloc((size_t)lo * 4 + 4)) || (rc = spos.alloc((size_t)lo * 4 + 4))) return rc;
      hipLaunchKernelGGL(dp_foldflag_kernel, dim3((lo + kBlock - 1) / kBlock), dim3(kBlock), 0, 0,
                         (const uint32_t *)fk2.p, 0, lo, (int32_t *)sflag.p);
      tb = 0;
      HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (const int32_t *)sflag.p, (int32_t *)spos.p, lo, (hipStream_t)0));
      if ((rc = tmp.need(tb))) return rc;
      tb = tmp.buf.bytes;
      HIPCHK(hipcub::DeviceScan::ExclusiveSum(tmp.buf.p, tb, (const int32_t *)sflag.p, (int32_t *)spos.p, lo, (hipStream_t)0));
      int32_t lp = 0, lf = 0;
      HIPCHK(hipMemcpy(&lp, (const int32_t *)spos.p + (lo - 1), 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(&lf, (const int32_t *)sflag.p + (lo - 1), 4, hipMemcpyDeviceToHost));
      const int ns = lp + lf;
      if ((rc = sstart.alloc(((size_t)ns + 2) * 4)) || (rc = srow.alloc((size_t)ns * 4 + 4))) return rc;
      hipLaunchKernelGGL(dp_foldstart_kernel, dim3((lo + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, 0,
                         (const int32_t *)sflag.p, (const int32_t *)spos.p, lo, (int32_t *)sstart.p);
      hipLaunchKernelGGL(dp_gatherkeys_kernel, dim3((ns + kBlock - 1) / kBlock), dim3(kBlock), 0, 0,
                         (const uint32_t *)fk2.p, (const int32_t *)sstart.p, ns, (int32_t *)srow.p);
      if ((rc = dl(m.P.send_ptr, sstart, (size_t)ns + 1)) || (rc = dl(m.P.send_row, srow, (size_t)ns))) return rc;
      if ((rc = m.send_ptr.upload(m.P.send_ptr.data(), m.P.send_ptr.size() * 4)) || (rc = m.send_idx.alloc((size_t)lo * 4 + 4)))
        return rc;
      HIPCHK(hipMemcpy(m.send_idx.p, fv2.p, (size_t)lo * 4, hipMemcpyDeviceToDevice));
      m.send_idx.bytes = (size_t)lo * 4;
    }
    if (F > 0) {
      DevBuf flag, pos, start, restlen, restoff;
      if ((rc = flag.alloc((size_t)F * 4 + 4)) || (rc = pos.alloc((size_t)F * 4 + 4))) return rc;
      hipLaunchKernelGGL(dp_foldflag_kernel, dim3((F + kBlock - 1) / kBlock), dim3(kBlock), 0, 0,
                         (const uint32_t *)fk2.p, lo, lo + F, (int32_t *)flag.p);
      tb = 0;
      HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (const int32_t *)flag.p, (int32_t *)pos.p, F, (hipStream_t)0));
      if ((rc = tmp.need(tb))) return rc;
      tb = tmp.buf.bytes;
      HIPCHK(hipcub::DeviceScan::ExclusiveSum(tmp.buf.p, tb, (const int32_t *)flag.p, (int32_t *)pos.p, F, (hipStream_t)0));
      int32_t lastp = 0, lastf = 0;
      HIPCHK(hipMemcpy(&lastp, (const int32_t *)pos.p + (F - 1), 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(&lastf, (const int32_t *)flag.p + (F - 1), 4, hipMemcpyDeviceToHost));
      nfold = lastp + lastf;
      if ((rc = start.alloc(((size_t)nfold + 2) * 4)) || (rc = restlen.alloc((size_t)nfold * 4 + 4)) ||
          (rc = restoff.alloc((size_t)nfold * 4 + 4)))
        return rc;
      hipLaunchKernelGGL(dp_foldstart_kernel, dim3((F + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, 0,
                         (const int32_t *)flag.p, (const int32_t *)pos.p, F, (int32_t *)start.p);
      hipLaunchKernelGGL(dp_foldlen_kernel, dim3((nfold + kBlock - 1) / kBlock), dim3(kBlock), 0, 0,
                         (const int32_t *)start.p, nfold, (int32_t *)restlen.p);
      tb = 0;
      HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (const int32_t *)restlen.p, (int32_t *)restoff.p, nfold,
                                              (hipStream_t)0));
      if ((rc = tmp.need(tb))) return rc;
      tb = tmp.buf.bytes;
      HIPCHK(hipcub::DeviceScan::ExclusiveSum(tmp.buf.p, tb, (const int32_t *)restlen.p, (int32_t *)restoff.p, nfold,
                                              (hipStream_t)0));
      int32_t ro = 0, rl = 0;
      HIPCHK(hipMemcpy(&ro, (const int32_t *)restoff.p + (nfold - 1), 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(&rl, (const int32_t *)restlen.p + (nfold - 1), 4, hipMemcpyDeviceToHost));
      const size_t nrest = (size_t)ro + rl;
      if ((rc = m.fold_rec.alloc(((size_t)nfold + 1) * sizeof(int4) + 64)) || (rc = m.fold_idx.alloc(nrest * 4 + 64)))
        return rc;
      m.fold_rec.bytes = ((size_t)nfold + 1) * sizeof(int4);
      m.fold_idx.bytes = nrest * 4;
      hipLaunchKernelGGL(dp_foldrec_kernel, dim3((nfold + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, 0,
                         (const uint32_t *)fk2.p, (const int32_t *)fv2.p, lo, (const int32_t *)start.p,
                         (const int32_t *)restoff.p, nfold, rb, re, perm.empty() ? nullptr : (const int32_t *)S->perm.p,
                         (int4 *)m.fold_rec.p, (int32_t *)m.fold_idx.p);
      HIPCHK(hipGetLastError());
      HIPCHK(hipDeviceSynchronize()); // the temporaries of this block go out of scope
    }
  }
  if (nfold == 0) { // the padding record the host builder also uploads
    const int4 pad = make_int4(0, 0, -1, -1);
    if ((rc = m.fold_rec.upload(&pad, sizeof pad)) || (rc = m.fold_idx.alloc(0))) return rc;
    m.fold_idx.bytes = 0;
  }
  pt.lap("device: fold index");

  // ---- the numbers, through the maps (what cfs_hip_sym_update_values_* does later) -----------------
  if ((rc = m.vals.alloc(vlen * sizeof(V))) || (rc = m.cvals.alloc(clen * sizeof(V))) ||
      (rc = m.diag.alloc(((size_t)nvr + 1) * sizeof(V))))
    return rc;
  auto scatter = [&](DevBuf &dst, DevBuf &map, size_t cnt) {
    if (!cnt) return;
    const int grid = (int)std::min<size_t>((cnt + 255) / 256, 256 * 16);
    hipLaunchKernelGGL((cfs_value_scatter_kernel<V>), dim3(grid), dim3(256), 0, 0, (V *)dst.p,
                       (const int32_t *)map.p, (const V *)in.lva.p, (long long)cnt);
  };
  scatter(m.vals, val_map, vlen);
  scatter(m.cvals, cval_map, clen);
  scatter(m.diag, diag_map, (size_t)nvr + 1);
  m.vals.bytes = vlen * sizeof(V);
  // the far sections (without far entries: padding only, the kernel's loops never enter them)
  const size_t flen = (size_t)farlen + 256;
  DevBuf fval_map;
  if ((rc = m.fvals.alloc(flen * sizeof(V))) || (rc = m.frows.alloc(flen * 2)) || (rc = m.fcols.alloc(flen * 4)) ||
      (rc = fval_map.alloc(flen * 4)))
    return rc;
  HIPCHK(hipMemsetAsync(m.frows.p, 0, flen * 2, 0));
  HIPCHK(hipMemsetAsync(m.fcols.p, 0, flen * 4, 0));
  HIPCHK(hipMemsetAsync(fval_map.p, 0xff, flen * 4, 0));
  if (FS.active) {
    const long long Kf = FS.kept;
    DevBuf PL, PU, upkey, upkey2, uppay, uppay2;
    if ((rc = PL.alloc(((size_t)rows + 1) * 4)) || (rc = PU.alloc(((size_t)rows + 1) * 4)) ||
        (rc = upkey.alloc((size_t)Kf * 8 + 8)) || (rc = upkey2.alloc((size_t)Kf * 8 + 8)) ||
        (rc = uppay.alloc((size_t)Kf * 4 + 4)) || (rc = uppay2.alloc((size_t)Kf * 4 + 4)))
      return rc;
    for (DevBuf *pp : {&PL, &PU}) { // exclusive prefixes of farL / farU (their last entry is 0)
      const int32_t *src = (const int32_t *)(pp == &PL ? FS.farL.p : FS.farU.p);
      size_t tb = 0;
      HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, src, (int32_t *)pp->p, rows + 1, (hipStream_t)0));
      if ((rc = tmp.need(tb))) return rc;
      tb = tmp.buf.bytes;
      HIPCHK(hipcub::DeviceScan::ExclusiveSum(tmp.buf.p, tb, src, (int32_t *)pp->p, rows + 1, (hipStream_t)0));
    }
    const int32_t *d_perm = perm.empty() ? nullptr : (const int32_t *)S->perm.p;
    const int gr = (rows + kBlock - 1) / kBlock;
    const unsigned gk = (unsigned)((Kf + kBlock - 1) / kBlock);
#define CFS_FAR_ARGS_LOW                                                                                          \
  rows, rb, (const Tile *)d_tiles.p, (const int32_t *)FS.tor.p, (const int32_t *)S->brp.p, (const int32_t *)S->bci.p, \
      S->bsrc(), (const uint8_t *)FS.far.p, (const int32_t *)PL.p, d_perm, (uint16_t *)m.frows.p,                 \
      (int32_t *)m.fcols.p, (int32_t *)fval_map.p, (uint64_t *)upkey.p, (int32_t *)uppay.p
    if (sizeof(V) == 8) hipLaunchKernelGGL((dp_farlow_kernel<8>), dim3(gr), dim3(kBlock), 0, 0, CFS_FAR_ARGS_LOW);
    else hipLaunchKernelGGL((dp_farlow_kernel<4>), dim3(gr), dim3(kBlock), 0, 0, CFS_FAR_ARGS_LOW);
#undef CFS_FAR_ARGS_LOW
    int end_bit = 32;
    while (end_bit < 64 && ((unsigned long long)rows >> (end_bit - 32)) != 0) end_bit++;
    size_t tb = 0;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const uint64_t *)upkey.p, (uint64_t *)upkey2.p,
                                              (const int32_t *)uppay.p, (int32_t *)uppay2.p, (int)Kf, 0, end_bit,
                                              (hipStream_t)0));
    if ((rc = tmp.need(tb))) return rc;
    tb = tmp.buf.bytes;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(tmp.buf.p, tb, (const uint64_t *)upkey.p, (uint64_t *)upkey2.p,
                                              (const int32_t *)uppay.p, (int32_t *)uppay2.p, (int)Kf, 0, end_bit,
                                              (hipStream_t)0));
#define CFS_FAR_ARGS_UP                                                                                          \
  Kf, rb, (const Tile *)d_tiles.p, (const int32_t *)FS.tor.p, (const uint64_t *)upkey2.p, (const int32_t *)uppay2.p, \
      (const int32_t *)PU.p, (uint16_t *)m.frows.p, (int32_t *)m.fcols.p, (int32_t *)fval_map.p
    if (sizeof(V) == 8) hipLaunchKernelGGL((dp_farup_kernel<8>), dim3(gk), dim3(kBlock), 0, 0, CFS_FAR_ARGS_UP);
    else hipLaunchKernelGGL((dp_farup_kernel<4>), dim3(gk), dim3(kBlock), 0, 0, CFS_FAR_ARGS_UP);
#undef CFS_FAR_ARGS_UP
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize()); // the temporaries of this block go out of scope
  }
  scatter(m.fvals, fval_map, flen);
  m.has_value_map = opt.keep_value_map;
  if (opt.keep_value_map) { // positions in the caller's values[], as cfs_hip_sym_update_values_* needs them
    auto conv = [&](DevBuf &map, size_t cnt) {
      const int grid = (int)std::min<size_t>((cnt + 255) / 256, 256 * 16);
      hipLaunchKernelGGL(dp_map_to_caller_kernel, dim3(grid), dim3(kBlock), 0, 0, (int32_t *)map.p, (long long)cnt,
                         (const int32_t *)in.lrp.p, in.row_hi - in.row_lo, in.row_lo, (const int32_t *)in.rowptr.p);
    };
    conv(val_map, vlen);
    conv(cval_map, clen);
    conv(diag_map, (size_t)nvr + 1);
    m.val_map = std::move(val_map);
    m.cval_map = std::move(cval_map);
    m.diag_map = std::move(diag_map);
    conv(fval_map, flen);
    m.fval_map = std::move(fval_map);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  { // tiles with their ny (set by dp_fill_kernel), the counters, a last look at the flags
    HIPCHK(hipMemcpy(tiles.data(), d_tiles.p, (size_t)T * sizeof(Tile), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(h_ctr, ctr.p, C_COUNT * 8, hipMemcpyDeviceToHost));
    int f[F_COUNT];
    if ((rc = read_flags(flags, f))) return rc;
    if (any_flag(f, why)) return kUseHost;
  }
  pt.lap("device: values");

  // ---- the host's view of the schedule (metadata only; the arrays live on the device) ----------
  P.n = n;
  P.nranks = nranks;
  P.rank = rank;
  if (row_splits_in) P.row_splits.assign(row_splits_in, row_splits_in + nranks + 1);
  else P.row_splits = {0, n};
  P.row_begin = rb;
  P.row_end = re;
  P.nnz_low = in.nnz_low;
  P.nnz_diag = in.nnz_diag;
  P.nnz_full = 2 * in.nnz_low + in.nnz_diag;
  P.max_slots = L.max_slots;
  P.block_threads = L.block;
  P.wg_per_cu = L.wg_per_cu;
  P.ngroups = L.ngroups;
  P.deterministic = opt.deterministic;
  P.mirrored = mirror;
  P.mirror_entries = S->mirror_entries;
  P.onesided_slots = onesided;
  P.nhalo = halo;
  P.nvrows = nvr;
  P.stream_len = off;
  P.slot_len = soff;
  P.coo_len = coo;
  P.far_len = farlen;
  P.far_entries = FS.active ? FS.kept : 0;
  P.far_candidates = opt.hyb ? FS.marked : (int64_t)h_ctr[C_FARCAND];
  P.chained_packets = (int64_t)h_ctr[C_CHAINED];
  P.lane_packets = (int64_t)h_ctr[C_LANEPK];
  P.tiles = tiles;
  P.group_ptr = S->group_ptr;
  int lds_slots = 64;
  for (auto &t : P.tiles) lds_slots = std::max(lds_slots, (int)t.nslots);
  P.lds_slots = (lds_slots + 63) / 64 * 64;
  P.group_first.assign(nc, Tile{});
  for (int g = 0; g < nc; g++)
    if (P.group_ptr[g] < P.group_ptr[g + 1]) P.group_first[g] = P.tiles[P.group_ptr[g]];
  if (!perm.empty()) P.perm = perm;
  compute_launch_order(L, opt, P.tiles, P.group_ptr, S->cost, rb, P.launch_order);
  P.fold_dst.assign((size_t)nfold, 0); // (sizes only: the records are on the device)
  P.send_counts.assign(nranks, 0);
  for (int r : P.send_row) { // owner of every row this shard sends sums to (cfs_plan::Builder::fold_index)
    const int owner = (int)(std::upper_bound(P.row_splits.begin(), P.row_splits.end(), r) - P.row_splits.begin()) - 1;
    P.send_counts[owner]++;
  }
  // hand the remaining arrays over
  m.tiles = std::move(d_tiles);
  m.tiles.bytes = (size_t)T * sizeof(Tile);
  if (cache && cache->rb == rb && cache->re == re) { // for tune()'s later builds of these rows
    std::shared_ptr<Kept<V>> keep;
    if (KP) keep = std::static_pointer_cast<Kept<V>>(cache->device_keep);
    else {
      keep = std::make_shared<Kept<V>>();
      keep->in = std::move(in_local);
      keep->mirror = mirror;
    }
    if (!K && have_clusters && SC_local.rows) { // a new clustered placement (whichever order won)
      keep->SC = std::move(SC_local);
      keep->has_sc = true;
      keep->nc = nc;
      keep->max_slots = L.max_slots;
      keep->cost_model = opt.cost_model;
      keep->perm = perm_all;
      keep->chunk = cchunk;
    }
    cache->device_keep = keep;
  }
  pt.lap("device: metadata");
  return 0; // (the caller adopts the schedule once this function's temporaries are gone)
}

} // namespace cfs_dev
