// cfs_plan.hpp -- host-side construction of the MI355X tile schedule for the
// symmetric (lower-triangle) SpMV.  Pure C++ (no HIP): it is what tune() does
// on the host before anything is uploaded, and it is unit-testable on a CPU
// box through cfs_hip_sym_plan_check_* (structure only, no SpMV arithmetic).
//
// What the reference does at this point (include/matrix/csr_matrix.tpp):
//   partition_by_nrows :403-435, conflict_free_aposteriori :1204-1639,
//   color_greedy :2009-2363 -- rows are split among T threads, the strict
//   lower triangle + diagonal are extracted per thread, 16-row blocks are
//   coloured so that no two threads update the same y element in a phase.
// What this build does instead (MI355X-first, not a translation):
//   * rows are cut into TILES of consecutive rows; one 64-lane wavefront lane
//     owns one row, so the row-side sum y_i needs no reduction at all;
//   * every column a tile touches gets a 16-bit SLOT in that tile's LDS window
//     (own rows first, then the sorted "halo" columns left of the tile), so
//     the gather x[col] and the transposed update y[col] += a*x[row] both hit
//     LDS (ds_read / ds_add), and the index stream shrinks from 4 to 2 bytes;
//   * rows of a tile are sorted by length and stored in 64-row SLICES as
//     jagged diagonals (lane l holds row l of the slice), the first
//     4*floor(min_len/4) diagonals in 4-wide PACKETS laid out for 16-byte
//     coalesced loads, the rest one diagonal at a time: zero padding entries;
//   * conflicts BETWEEN tiles (the reference's direct conflicts, :1443-1451)
//     are not coloured away but deferred: a tile stores its halo sums to a
//     private strip with plain coalesced stores and a tiny second kernel folds
//     the strips into y through an inverted index in a fixed order (no global
//     atomics, no barriers between colours);
//   * tiles are dealt to persistent workgroups in contiguous, cost-balanced
//     groups, groups of neighbouring rows on the same XCD (blockIdx % 8).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace cfs_plan {

struct Options {
  int max_slots = 0;
  int max_tile_nnz = 0;
  int block_threads = 0;
  int flags = 0;
};

// one tile = one pass of a workgroup through prologue / slices / epilogue
struct Tile {
  int32_t row0;       // first own row (global index)
  int32_t nown;       // own rows = own slots [0, nown)
  int32_t nslots;     // nown + halo slots
  int32_t nslices;    // ceil(nown / 64)
  int32_t halo_off;   // offset of this tile's halo in halo_col[] and strip[]
  int32_t slice_base; // offset of this tile's slices in slice_off[]
  int64_t nnz_off;    // offset of this tile's entries in vals[] / slots[]
};
static_assert(sizeof(Tile) == 32, "Tile must stay 32 bytes");

constexpr int kLanes = 64;
constexpr int kPacket = 4;            // diagonals per packet
constexpr int kAlignEntries = 8;      // slice streams start on 8-entry bounds
constexpr int kMaxSlotsHard = 10240;  // 16 B/slot fp64 -> 160 KiB LDS

template <typename V> struct SymPlan {
  // problem
  int n = 0, row_begin = 0, row_end = 0, nranks = 1, rank = 0;
  std::vector<int> row_splits;
  int64_t nnz_low = 0, nnz_diag = 0, nnz_full = 0;
  // knobs actually used
  int max_slots = 0, block_threads = 0, ngroups = 0, lds_slots = 0;
  // schedule
  std::vector<Tile> tiles;
  std::vector<int32_t> group_ptr;   // [ngroups+1] tiles of persistent group g
  std::vector<int32_t> halo_col;    // [H] global column of every halo slot
  std::vector<uint32_t> rowinfo;    // [rows] sorted position -> local_row | len<<16
  std::vector<V> diag;              // [rows] diagonal, in sorted position order
  std::vector<uint32_t> slice_off;  // [S] entry offset of a slice inside its tile
  std::vector<V> vals;              // [stream_len]
  std::vector<uint16_t> slots;      // [stream_len]
  // halo fold (destinations inside [row_begin,row_end)), local row indices
  std::vector<int32_t> fold_row, fold_ptr, fold_idx;
  // remote contributions (destinations < row_begin), global row indices
  std::vector<int32_t> send_row, send_ptr, send_idx, send_counts;
  // receive side (filled by set_recv)
  int nrecv = 0;
  std::vector<int32_t> rfold_row, rfold_ptr, rfold_idx;
  int64_t stream_len = 0;
  std::string error;
};

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// Position of entry (packet g, lane l, diagonal j of the packet) inside a
// packet's 256 entries.  Values: two fully contiguous 16-byte-per-lane loads
// for fp64 ([half][lane][2]), one for fp32 ([lane][4]).  Slots: [lane][4].
template <typename V> inline int packet_val_pos(int l, int j) {
  if (sizeof(V) == 8) return (j >> 1) * 128 + l * 2 + (j & 1);
  return l * 4 + j;
}
inline int packet_slot_pos(int l, int j) { return l * 4 + j; }

// nnz_low-balanced shard boundaries at multiples of 16 rows (BlkFactor,
// csr_matrix.tpp:418; the reference's partition_by_nrows balances rows, which
// is poor for a lower triangle -- SURVEY.md 8e).
inline void balanced_splits(int n, const int *rowptr, const int *colind,
                            int nranks, int *row_splits) {
  std::vector<int64_t> low(n + 1, 0);
  for (int i = 0; i < n; i++) {
    int c = 0;
    for (int j = rowptr[i]; j < rowptr[i + 1]; j++)
      if (colind[j] < i) c++;
    low[i + 1] = low[i] + c + 3; // +3: per-row traffic (rowinfo, diag, x, y)
  }
  row_splits[0] = 0;
  for (int r = 1; r < nranks; r++) {
    int64_t target = low[n] * r / nranks;
    int i = (int)(std::lower_bound(low.begin(), low.end(), target) - low.begin());
    i = (i + 8) / 16 * 16;
    if (i > n) i = n;
    if (i < row_splits[r - 1]) i = row_splits[r - 1];
    row_splits[r] = i;
  }
  row_splits[nranks] = n;
}

// Build the plan for rows [row_splits[rank], row_splits[rank+1]) of the full
// CSR.  Returns false (plan.error set) when the matrix cannot be scheduled.
template <typename V>
bool build_plan(int n, const int *rowptr, const int *colind, const V *values,
                int nranks, int rank, const int *row_splits_in,
                const Options &opt, SymPlan<V> &P) {
  P = SymPlan<V>();
  P.n = n;
  P.nranks = nranks;
  P.rank = rank;
  if (row_splits_in)
    P.row_splits.assign(row_splits_in, row_splits_in + nranks + 1);
  else
    P.row_splits = {0, n};
  const int rb = P.row_begin = P.row_splits[rank];
  const int re = P.row_end = P.row_splits[rank + 1];
  if (rb < 0 || re > n || rb > re) {
    P.error = "bad row_splits";
    return false;
  }
  const int rows = re - rb;
  const int slot_bytes = 2 * (int)sizeof(V);
  const int hard_slots = 160 * 1024 / slot_bytes;
  int max_slots = opt.max_slots > 0 ? opt.max_slots : 2560;
  if (max_slots > hard_slots) max_slots = hard_slots;
  if (max_slots > 65536) max_slots = 65536;
  if (max_slots < 64) max_slots = 64;
  P.max_slots = max_slots;
  int block = opt.block_threads > 0 ? opt.block_threads : 256;
  if (block != 256 && block != 512 && block != 1024) {
    P.error = "block_threads must be 256, 512 or 1024";
    return false;
  }
  P.block_threads = block;
  const int64_t max_tile_nnz =
      opt.max_tile_nnz > 0 ? opt.max_tile_nnz : (int64_t)1 << 30;

  // ---- lower counts -----------------------------------------------------
  std::vector<int32_t> lcnt(rows, 0);
  int64_t nnz_low = 0, nnz_diag = 0;
#pragma omp parallel for schedule(static) reduction(+ : nnz_low, nnz_diag)
  for (int i = rb; i < re; i++) {
    int c = 0;
    for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
      if (colind[j] < i) c++;
      else if (colind[j] == i) nnz_diag++;
    }
    lcnt[i - rb] = c;
    nnz_low += c;
  }
  P.nnz_low = nnz_low;
  P.nnz_diag = nnz_diag;
  P.nnz_full = 2 * nnz_low + nnz_diag;

  // ---- cut rows into tiles (sequential, O(nnz)) --------------------------
  {
    std::vector<int32_t> stamp(n > 0 ? n : 1, -1);
    int row = rb, tid = 0;
    while (row < re) {
      Tile t{};
      t.row0 = row;
      int nown = 0, nhalo = 0;
      int64_t nnz = 0;
      while (row < re) {
        int newh = 0, len = 0;
        for (int j = rowptr[row]; j < rowptr[row + 1]; j++) {
          int c = colind[j];
          if (c >= row) continue;
          len++;
          if (c < t.row0 && stamp[c] != tid) {
            stamp[c] = tid;
            newh++;
          }
        }
        if (len > 65535) {
          P.error = "row with more than 65535 lower entries";
          return false;
        }
        bool fits = (nown + 1 + nhalo + newh <= max_slots) &&
                    (nnz + len <= max_tile_nnz || nown == 0) && nown < 65535;
        if (!fits) {
          if (nown == 0) {
            P.error = "a single row needs more LDS slots than max_slots "
                      "(dense row): unsupported by the tile schedule";
            return false;
          }
          break; // stale stamps are harmless: the next tile uses tid + 1
        }
        nown++;
        nhalo += newh;
        nnz += len;
        row++;
      }
      t.nown = nown;
      t.nslots = nown + nhalo;
      t.nslices = (nown + kLanes - 1) / kLanes;
      P.tiles.push_back(t);
      tid++;
    }
  }
  const int T = (int)P.tiles.size();

  // ---- offsets -------------------------------------------------------------
  {
    int64_t halo = 0, slices = 0;
    for (auto &t : P.tiles) {
      t.halo_off = (int32_t)halo;
      t.slice_base = (int32_t)slices;
      halo += t.nslots - t.nown;
      slices += t.nslices;
      if (halo > 0x7fffffffLL) {
        P.error = "halo index overflow";
        return false;
      }
    }
    P.halo_col.assign((size_t)halo, 0);
    P.slice_off.assign((size_t)slices, 0);
    P.rowinfo.assign((size_t)rows, 0);
    P.diag.assign((size_t)rows, V(0));
  }

  // ---- per tile: halo map, length sort, stream sizes ------------------------
  // pass A computes per-tile stream length and slice offsets, pass B fills.
  std::vector<int64_t> tile_len(T, 0);
  std::vector<std::vector<int32_t>> perm_store; // not kept: recomputed in pass B
  (void)perm_store;
  auto sort_rows = [&](const Tile &t, std::vector<int32_t> &perm) {
    // stable counting sort of local rows by lower length, descending
    perm.resize(t.nown);
    int maxlen = 0;
    for (int r = 0; r < t.nown; r++)
      maxlen = std::max(maxlen, (int)lcnt[t.row0 - rb + r]);
    std::vector<int32_t> cnt(maxlen + 2, 0);
    for (int r = 0; r < t.nown; r++) cnt[maxlen - lcnt[t.row0 - rb + r] + 1]++;
    for (int k = 0; k <= maxlen; k++) cnt[k + 1] += cnt[k];
    for (int r = 0; r < t.nown; r++) perm[cnt[maxlen - lcnt[t.row0 - rb + r]]++] = r;
  };
#pragma omp parallel for schedule(dynamic, 8)
  for (int ti = 0; ti < T; ti++) {
    const Tile &t = P.tiles[ti];
    std::vector<int32_t> perm;
    sort_rows(t, perm);
    int64_t off = 0;
    for (int s = 0; s < t.nslices; s++) {
      off = align_up(off, kAlignEntries);
      P.slice_off[t.slice_base + s] = (uint32_t)off;
      int p0 = s * kLanes, p1 = std::min(p0 + kLanes, (int)t.nown);
      for (int p = p0; p < p1; p++) off += lcnt[t.row0 - rb + perm[p]];
    }
    tile_len[ti] = align_up(off, kAlignEntries);
  }
  {
    int64_t off = 0;
    for (int ti = 0; ti < T; ti++) {
      P.tiles[ti].nnz_off = off;
      off += tile_len[ti];
    }
    P.stream_len = off;
    P.vals.assign((size_t)off, V(0));
    P.slots.assign((size_t)off, 0);
  }
  bool dup_error = false;
#pragma omp parallel
  {
    std::vector<int32_t> colmap(n > 0 ? n : 1, -1); // col -> halo slot (per thread)
    std::vector<int32_t> perm, hcols;
#pragma omp for schedule(dynamic, 8)
    for (int ti = 0; ti < T; ti++) {
      const Tile &t = P.tiles[ti];
      // halo: unique columns < row0, ascending
      hcols.clear();
      for (int r = 0; r < t.nown; r++) {
        int i = t.row0 + r;
        for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
          int c = colind[j];
          if (c < t.row0 && colmap[c] < 0) {
            colmap[c] = 0;
            hcols.push_back(c);
          }
        }
      }
      std::sort(hcols.begin(), hcols.end());
      if ((int)hcols.size() != t.nslots - t.nown) dup_error = true;
      for (size_t h = 0; h < hcols.size(); h++) {
        colmap[hcols[h]] = t.nown + (int)h;
        P.halo_col[t.halo_off + h] = hcols[h];
      }
      sort_rows(t, perm);
      V *tv = P.vals.data() + t.nnz_off;
      uint16_t *ts = P.slots.data() + t.nnz_off;
      std::vector<int32_t> start(kLanes), len(kLanes);
      for (int s = 0; s < t.nslices; s++) {
        int p0 = s * kLanes, m = std::min(kLanes, (int)t.nown - p0);
        for (int l = 0; l < m; l++) {
          int r = perm[p0 + l], i = t.row0 + r;
          len[l] = lcnt[i - rb];
          P.rowinfo[t.row0 - rb + p0 + l] = (uint32_t)r | ((uint32_t)len[l] << 16);
          // first lower entry of row i; the lower entries are a prefix of the
          // row because columns ascend (csr_matrix.tpp:74-107)
          start[l] = rowptr[i];
          V d = V(0);
          for (int j = rowptr[i]; j < rowptr[i + 1]; j++)
            if (colind[j] == i) d = values[j]; // last duplicate wins, like :1292
          P.diag[t.row0 - rb + p0 + l] = d;
        }
        auto entry = [&](int l, int k, V &v, uint16_t &sl) {
          // k-th lower entry of the row held by lane l.  Rows are scanned
          // rather than assumed prefix-ordered, so unsorted input still works.
          int i = t.row0 + perm[p0 + l], seen = 0;
          for (int j = start[l]; j < rowptr[i + 1]; j++) {
            if (colind[j] >= i) continue;
            if (seen == k) {
              int c = colind[j];
              v = values[j];
              sl = (uint16_t)(c >= t.row0 ? c - t.row0 : colmap[c]);
              return;
            }
            seen++;
          }
        };
        int64_t base = P.slice_off[t.slice_base + s];
        int minlen = (m == kLanes) ? len[kLanes - 1] : 0;
        int nfull = minlen / kPacket;
        // fast path: columns ascend, so the k-th lower entry is rowptr[i]+k
        bool prefix = true;
        for (int l = 0; l < m && prefix; l++) {
          int i = t.row0 + perm[p0 + l];
          for (int k = 0; k < len[l]; k++)
            if (colind[rowptr[i] + k] >= i) {
              prefix = false;
              break;
            }
        }
        auto fetch = [&](int l, int k, V &v, uint16_t &sl) {
          if (prefix) {
            int i = t.row0 + perm[p0 + l];
            int j = rowptr[i] + k, c = colind[j];
            v = values[j];
            sl = (uint16_t)(c >= t.row0 ? c - t.row0 : colmap[c]);
          } else {
            entry(l, k, v, sl);
          }
        };
        for (int g = 0; g < nfull; g++)
          for (int l = 0; l < kLanes; l++)
            for (int j = 0; j < kPacket; j++) {
              V v;
              uint16_t sl;
              fetch(l, g * kPacket + j, v, sl);
              tv[base + (int64_t)g * 256 + packet_val_pos<V>(l, j)] = v;
              ts[base + (int64_t)g * 256 + packet_slot_pos(l, j)] = sl;
            }
        int64_t o = base + (int64_t)nfull * 256;
        int maxlen = m > 0 ? len[0] : 0;
        for (int k = nfull * kPacket; k < maxlen; k++)
          for (int l = 0; l < m && len[l] > k; l++) {
            V v;
            uint16_t sl;
            fetch(l, k, v, sl);
            tv[o] = v;
            ts[o] = sl;
            o++;
          }
      }
      for (int c : hcols) colmap[c] = -1;
    }
  }
  if (dup_error) {
    P.error = "internal: halo count mismatch";
    return false;
  }

  // ---- halo fold index: strips -> destination rows, fixed order --------------
  {
    const int64_t H = (int64_t)P.halo_col.size();
    std::vector<int32_t> lcount(rows + 1, 0), rcount(rb + 1, 0);
    for (int64_t q = 0; q < H; q++) {
      int c = P.halo_col[q];
      if (c >= rb) lcount[c - rb + 1]++;
      else rcount[c + 1]++;
    }
    // compact destination lists
    std::vector<int32_t> lpos(rows, -1), rpos(rb > 0 ? rb : 1, -1);
    P.fold_ptr.push_back(0);
    for (int r = 0; r < rows; r++)
      if (lcount[r + 1]) {
        lpos[r] = (int)P.fold_row.size();
        P.fold_row.push_back(r);
        P.fold_ptr.push_back(P.fold_ptr.back() + lcount[r + 1]);
      }
    P.send_ptr.push_back(0);
    for (int r = 0; r < rb; r++)
      if (rcount[r + 1]) {
        rpos[r] = (int)P.send_row.size();
        P.send_row.push_back(r);
        P.send_ptr.push_back(P.send_ptr.back() + rcount[r + 1]);
      }
    P.fold_idx.assign((size_t)P.fold_ptr.back(), 0);
    P.send_idx.assign((size_t)P.send_ptr.back(), 0);
    std::vector<int32_t> lfill(P.fold_ptr.begin(), P.fold_ptr.end() - 1);
    std::vector<int32_t> rfill(P.send_ptr.begin(), P.send_ptr.end() - 1);
    for (int64_t q = 0; q < H; q++) { // ascending strip index => tile order
      int c = P.halo_col[q];
      if (c >= rb) P.fold_idx[lfill[lpos[c - rb]]++] = (int32_t)q;
      else P.send_idx[rfill[rpos[c]]++] = (int32_t)q;
    }
    P.send_counts.assign(nranks, 0);
    for (int r : P.send_row) {
      int owner = (int)(std::upper_bound(P.row_splits.begin(), P.row_splits.end(), r) -
                        P.row_splits.begin()) - 1;
      P.send_counts[owner]++;
    }
  }

  // ---- persistent groups: contiguous tiles, cost balanced ---------------------
  {
    int lds_slots = 64;
    for (auto &t : P.tiles) lds_slots = std::max(lds_slots, (int)t.nslots);
    P.lds_slots = (lds_slots + 63) / 64 * 64;
    int64_t lds_bytes = (int64_t)P.lds_slots * slot_bytes;
    int wg_per_cu = (int)std::min<int64_t>(160 * 1024 / lds_bytes, 2048 / block);
    if (wg_per_cu < 1) wg_per_cu = 1;
    int ngroups = 256 * wg_per_cu;
    if (ngroups > T) ngroups = (T + 7) / 8 * 8;
    if (ngroups < 8) ngroups = 8;
    P.ngroups = ngroups;
    std::vector<int64_t> cost(T + 1, 0);
    for (int ti = 0; ti < T; ti++) {
      const Tile &t = P.tiles[ti];
      int64_t c = tile_len[ti] * (int64_t)(sizeof(V) + 2) +
                  (int64_t)t.nown * (4 + 3 * sizeof(V)) +
                  (int64_t)(t.nslots - t.nown) * (4 + 2 * sizeof(V)) + 512;
      cost[ti + 1] = cost[ti] + c;
    }
    P.group_ptr.assign(ngroups + 1, T);
    P.group_ptr[0] = 0;
    for (int g = 1; g < ngroups; g++) {
      int64_t target = cost[T] * g / ngroups;
      int ti = (int)(std::lower_bound(cost.begin(), cost.end(), target) - cost.begin());
      // nearest boundary
      if (ti > 0 && target - cost[ti - 1] < cost[ti] - target) ti--;
      if (ti < P.group_ptr[g - 1]) ti = P.group_ptr[g - 1];
      if (ti > T) ti = T;
      P.group_ptr[g] = ti;
    }
    P.group_ptr[ngroups] = T;
  }
  return true;
}

// Receive side of a shard: recv_rows[k] is the global row of the k-th value of
// the receive buffer.  Builds rfold_* (fixed order: ascending buffer index).
template <typename V>
bool set_recv(SymPlan<V> &P, int nrecv, const int *recv_rows) {
  const int rb = P.row_begin, rows = P.row_end - P.row_begin;
  std::vector<int32_t> cnt(rows + 1, 0);
  for (int k = 0; k < nrecv; k++) {
    int r = recv_rows[k] - rb;
    if (r < 0 || r >= rows) {
      P.error = "recv row outside this shard";
      return false;
    }
    cnt[r + 1]++;
  }
  P.nrecv = nrecv;
  P.rfold_row.clear();
  P.rfold_ptr.assign(1, 0);
  std::vector<int32_t> pos(rows > 0 ? rows : 1, -1);
  for (int r = 0; r < rows; r++)
    if (cnt[r + 1]) {
      pos[r] = (int)P.rfold_row.size();
      P.rfold_row.push_back(r);
      P.rfold_ptr.push_back(P.rfold_ptr.back() + cnt[r + 1]);
    }
  P.rfold_idx.assign((size_t)nrecv, 0);
  std::vector<int32_t> fill(P.rfold_ptr.begin(), P.rfold_ptr.end() - 1);
  for (int k = 0; k < nrecv; k++) P.rfold_idx[fill[pos[recv_rows[k] - rb]]++] = k;
  return true;
}

// Decode the device format back into (row, col, value) triples of the strict
// lower triangle, walking it exactly as the kernel does (packets, jagged
// diagonals, slot -> column through the own range / halo map).  Structure
// check for the CPU test-suite; performs no SpMV arithmetic.
template <typename V>
void decode_plan(const SymPlan<V> &P, std::vector<int32_t> &row,
                 std::vector<int32_t> &col, std::vector<V> &val) {
  row.clear();
  col.clear();
  val.clear();
  const int rb = P.row_begin;
  for (const Tile &t : P.tiles) {
    const V *tv = P.vals.data() + t.nnz_off;
    const uint16_t *ts = P.slots.data() + t.nnz_off;
    auto slot_col = [&](int s) {
      return s < t.nown ? t.row0 + s : P.halo_col[t.halo_off + (s - t.nown)];
    };
    for (int s = 0; s < t.nslices; s++) {
      int p0 = s * kLanes, m = std::min(kLanes, (int)t.nown - p0);
      int len[kLanes], r[kLanes];
      for (int l = 0; l < kLanes; l++) {
        uint32_t info = l < m ? P.rowinfo[t.row0 - rb + p0 + l] : 0;
        r[l] = info & 0xffff;
        len[l] = info >> 16;
      }
      int64_t base = P.slice_off[t.slice_base + s];
      int minlen = len[kLanes - 1], nfull = minlen / kPacket;
      for (int g = 0; g < nfull; g++)
        for (int l = 0; l < kLanes; l++)
          for (int j = 0; j < kPacket; j++) {
            row.push_back(t.row0 + r[l]);
            col.push_back(slot_col(ts[base + (int64_t)g * 256 + packet_slot_pos(l, j)]));
            val.push_back(tv[base + (int64_t)g * 256 + packet_val_pos<V>(l, j)]);
          }
      int64_t o = base + (int64_t)nfull * 256;
      for (int k = nfull * kPacket; k < len[0]; k++)
        for (int l = 0; l < kLanes && len[l] > k; l++) {
          row.push_back(t.row0 + r[l]);
          col.push_back(slot_col(ts[o]));
          val.push_back(tv[o]);
          o++;
        }
    }
  }
}

} // namespace cfs_plan
