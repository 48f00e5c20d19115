// cfs_plan.hpp -- host-side construction of the MI355X tile schedule for the
// symmetric (lower-triangle) SpMV.  Pure C++ (no HIP): it is what tune() does
// on the host before anything is uploaded, and it is unit-testable on a CPU
// box through cfs_hip_sym_plan_check_* (structure only, no SpMV arithmetic).
//
// What the reference does at this point (include/matrix/csr_matrix.tpp):
//   partition_by_nrows :403-435, conflict_free_aposteriori :1204-1639,
//   color_greedy :2009-2363, split_by_bandwidth :312-401 -- rows are split among
//   T threads, the strict lower triangle + diagonal are extracted per thread,
//   16-row blocks are coloured so that no two threads update the same y element
//   in a phase; HYB moves entries far from the diagonal to an unsymmetric side CSR.
// What this build does instead (MI355X-first, not a translation):
//   * rows are cut into TILES of consecutive rows; one 64-lane wavefront lane
//     owns one row, so the row-side sum y_i needs no reduction at all;
//   * every column a tile touches gets a 16-bit SLOT in that tile's LDS window
//     (own rows first, then the sorted "halo" columns left of the tile), so
//     the gather x[col] and the transposed update y[col] += a*x[row] both hit
//     LDS (ds_read / ds_add), and the index stream shrinks from 4 to 2 bytes;
//   * rows of a tile are stored in 64-row SLICES (lane l holds row l of the
//     slice; siblings -- the rows of one mesh node -- stay in one slice, slices
//     are sorted by length inside) as a uniform stream of 4-diagonal PACKETS
//     laid out for 16-byte coalesced loads; a packet covers the lanes whose
//     row still has 4 more entries (a prefix of the lanes), so there is no
//     padding and no special case in the stream.  A lane whose column sequence
//     is a prefix of an earlier lane's stores no slots and reads that lane's.
//     The len%4 last entries of every row go to a small per-tile COO section
//     (row slot, column slot, value) handled with LDS atomics on both sides;
//   * conflicts BETWEEN tiles (the reference's direct conflicts, :1443-1451)
//     are not coloured away but deferred: a tile stores its halo sums to a
//     private strip with plain coalesced stores and a tiny second kernel folds
//     the strips into y through an inverted index in a fixed order (no global
//     atomics, no barriers between colours);
//   * FAR entries (Format::hyb, the reference's split_by_bandwidth idea): a halo
//     column that a tile uses only once costs a slot, a slot-table entry, an x
//     gather, a strip store and a fold entry for ONE nonzero.  Such an entry leaves
//     the symmetric tile format: it is stored by BOTH tiles it touches, each as
//     (value, own row slot, global column), and processed one-sided -- y_l[row] +=
//     a * x[col] with x gathered from global memory: no slot, no strip, no fold;
//   * tiles are dealt to persistent workgroups in contiguous, cost-balanced
//     groups, groups of neighbouring rows on the same XCD (blockIdx % 8);
//   * a 1-D row block of a sharded matrix stores its off-block entries
//     MIRRORED (both ranks keep them, each processes its side), so that no
//     contribution leaves the rank and a sharded SpMV needs no exchange.
#pragma once

#include <omp.h>
#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace cfs_plan {

struct Options {
  int max_slots = 0;
  int max_tile_nnz = 0;
  int block_threads = 0;
  int flags = 0;
  bool reorder = true; // cluster rows so that tiles are compact in every direction
  int force_order = 0; // 0 = pick the order with fewer halo slots, 1 = natural, 2 = clustered
  // relative work share of every persistent group (size = number of groups,
  // any positive scale); empty = equal shares.
  std::vector<double> group_share;
  bool cost_model = true; // tiles / halo / dispatch wave in the cost of a group (ClusterCost)
  bool combine_siblings = true; // leadlane bits 6 / 7: one y-window atomic per run of sibling lanes
  // Shards (nranks > 1): off-block entries are stored by BOTH ranks they touch and
  // processed one-sided (row side only): a row of the block also carries its
  // entries a_ri of rows r owned by higher ranks, and its entries left of the
  // block no longer update y of the lower rank.  No contribution ever leaves
  // the rank, so an SpMV needs no exchange at all (x is replicated anyway).
  // false = the exchange form: contributions to lower ranks are packed and sent.
  bool mirror_offblock = true;
  int wg_per_cu = 0; // measured residency of the tile kernel (0 = estimate)
  int num_cus = 0;   // compute units of the device (0 = 256, MI355X)
  // HYB: entries whose halo column is used at most `far_uses` times by their tile
  // become FAR entries (see the header comment)
  bool hyb = false;
  int far_uses = 1;
  bool count_far = false; // HYB off: still report SymPlan::far_candidates
  // bit-reproducible results: the y window takes two 8-byte integer words per slot
  // (fixed-point sums, integer LDS atomics) instead of one fp64 word; no far entries
  bool deterministic = false;
  // deterministic build: per ORIGINAL row an exponent e with 2^e > its 1-norm sum_j |a_ij| (the
  // whole row, both triangles): the scale of the fixed-point sums of that row's slot, so that
  // their precision does not depend on how the matrix is scaled.  Set by build_plan.
  const int16_t *row_exp = nullptr;
  // also record, for every stored value of the device format, its position in the caller's
  // CSR value array: new values of the same sparsity pattern can then be poured into the
  // existing schedule by a device kernel (cfs_hip_sym_update_values_*) instead of tune()
  bool keep_value_map = false;
};

// one tile = one pass of a workgroup through prologue / slices / epilogue
struct Tile {
  int32_t row0;       // first own row (global index, schedule space)
  int32_t nown;       // own rows = own slots [0, nown)
  int32_t nslots;     // nown + halo slots
  int32_t nvrows;     // virtual rows (>= nown: long rows are split over lanes)
  int32_t halo_off;   // offset of this tile's halo in halo_col[] and strip[]
  int32_t slice_base; // offset of this tile's slices in slice_meta[]
  int64_t nnz_off;    // offset of this tile's value stream in vals[]
  int64_t sl_off;     // offset of this tile's slot stream in slots[]
  int32_t coo_off;    // offset of this tile's COO section in cvals/crows/ccols
  int32_t ncoo;       // leftover entries (len % 4 per row)
  int32_t slot_off;   // offset of this tile's slots in slot_col[]
  int32_t vrow_off;   // offset of this tile's virtual rows in rowinfo[] / diag[]
  int32_t ny;         // slots with a y window entry: own rows + in-block halo; slots
                      // [ny, nslots) are off-block columns of a mirrored shard (x only)
  int32_t far_off;    // offset of this tile's FAR section in fvals/frows/fcols
  int32_t nfar;       // far entries: [0, nfar_low) belong to own rows' lower triangle,
  int32_t nfar_low;   // [nfar_low, nfar) are the mirror images of other tiles' far entries
  int32_t nslices;    // ceil(nvrows / 64)
  int32_t aexp;       // every stored value of the tile (diagonal included) is < 2^aexp in
                      // magnitude: the scale of the deterministic build's fixed-point sums
};
static_assert(sizeof(Tile) == 80, "Tile must stay 80 bytes");

struct SliceMeta {
  uint32_t voff;      // entry offset of the slice's first packet in the tile's value stream
  uint32_t soff_cnt0; // bits 0..24: offset in the tile's slot stream; bits 25..31: lanes of packet 0
  uint64_t leaders;   // bit l = lane l stores its own slots (a LEADER); the others read the
                      // slots of the leader lane named in SymPlan::leadlane
};
static_assert(sizeof(SliceMeta) == 16, "SliceMeta must stay 16 bytes");

constexpr int kLanes = 64;
constexpr int kPacket = 4;            // diagonals per packet
constexpr int kAlignEntries = 8;      // slice streams start on 8-entry bounds
constexpr int kDefaultSlots = 4992;    // 2 workgroups x (4992 x 16 B + 16 B) fit the 160 KiB of a CU
constexpr int kDefaultBlock = 512;    // 8 waves per workgroup, 16 per CU (measured best)
constexpr int kStaticLds = 16;        // slice ticket counter
constexpr int kAexpNonFinite = 30000; // Tile::aexp of a tile that holds a NaN / Inf value
constexpr int16_t kExpNonFinite = 32000; // SymPlan::slot_exp of a row that holds a NaN / Inf value
constexpr int kSlotsPerThread = 10;   // LDS slots one thread fills/flushes (registers)
constexpr int kStreamPad = 256;       // padding entries behind the value / slot streams (one packet)

// Threads for the host-side schedule build: the OpenMP default, capped by the
// affinity mask and by the cgroup CPU quota (a container that shows 256 cores but
// grants 16 CPUs of time makes 256 spinning threads many times slower than 16;
// measured on the MI355X boxes).  CFS_HOST_THREADS overrides.
inline int host_threads() {
  static int cached = 0;
  if (cached > 0) return cached;
  int t = omp_get_max_threads();
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0)
    t = std::min(t, (int)CPU_COUNT(&set));
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64];
    long period = 0;
    if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      const long quota = atol(q);
      if (quota > 0) t = std::min<long>(t, (quota + period - 1) / period);
    }
    fclose(f);
  }
  if (const char *e = getenv("CFS_HOST_THREADS"))
    if (atoi(e) > 0) t = atoi(e);
  cached = std::max(1, t);
  return cached;
}

// The arrays that hold one entry per stored nonzero (hundreds of MB): a vector
// whose resize() does NOT touch the memory, filled by all host threads.  A
// std::vector would zero 0.7 GB of fresh pages from one thread first -- half of
// tune()'s host time went into those page faults.
template <typename T> struct NoInitAlloc : std::allocator<T> {
  template <typename U> struct rebind {
    using other = NoInitAlloc<U>;
  };
  NoInitAlloc() = default;
  template <typename U> NoInitAlloc(const NoInitAlloc<U> &) {}
  template <typename U> void construct(U *p) { ::new ((void *)p) U; } // default-init: no store
  template <typename U, typename A0, typename... A> void construct(U *p, A0 &&a0, A &&...a) {
    ::new ((void *)p) U(std::forward<A0>(a0), std::forward<A>(a)...);
  }
};
template <typename T> using BigVec = std::vector<T, NoInitAlloc<T>>;
template <typename T> inline void release(BigVec<T> &v) { BigVec<T>().swap(v); }
// ... and handing hundreds of MB back to the kernel takes tens of milliseconds: a
// big array that is no longer needed is freed by a thread of its own
template <typename T> inline void release_async(BigVec<T> &v) {
  if (v.capacity() * sizeof(T) < ((size_t)64 << 20)) return release(v);
  auto *h = new BigVec<T>();
  h->swap(v);
  try {
    std::thread([h] { delete h; }).detach();
  } catch (...) { // no thread to be had: free it here
    delete h;
  }
}
// v = n copies of val, pages first touched (and filled) by all host threads
template <typename T> inline void par_assign(BigVec<T> &v, size_t n, T val) {
  v.clear();
  v.resize(n);
  T *p = v.data();
  const size_t chunk = (size_t)1 << 18;
  const int64_t nch = (int64_t)((n + chunk - 1) / chunk);
#pragma omp parallel for schedule(static) num_threads(host_threads())
  for (int64_t c = 0; c < nch; c++)
    std::fill(p + (size_t)c * chunk, p + std::min(n, (size_t)(c + 1) * chunk), val);
}

template <typename V> struct SymPlan {
  // problem
  int n = 0, row_begin = 0, row_end = 0, nranks = 1, rank = 0;
  std::vector<int> row_splits;
  int64_t nnz_low = 0, nnz_diag = 0, nnz_full = 0;
  // knobs actually used
  int max_slots = 0, block_threads = 0, lds_slots = 0;
  int wg_per_cu = 1; // co-resident workgroups per CU the persistent grid was sized for
  bool deterministic = false;
  // launch shape: `ngroups` persistent workgroups (a multiple of 8); workgroup b runs
  // group (b % 8) * (ngroups / 8) + b / 8
  int ngroups = 0;
  // schedule
  std::vector<Tile> tiles;
  std::vector<int32_t> group_ptr;   // [ngroups+1] tiles of persistent group g
  std::vector<Tile> group_first;    // [ngroups] copy of each group's first tile: one
                                    // dependent load less at kernel start
  std::vector<int32_t> launch_order; // [ngroups] launch slot -> group (workgroup b runs slot
                                    // (b % 8) * (ngroups / 8) + b / 8); a permutation inside
                                    // every XCD's run of ngroups / 8 slots
  std::vector<int32_t> halo_col;    // [H] column of every halo slot (schedule space)
  std::vector<int32_t> slot_col;    // [sum nslots + 1] ORIGINAL column of every slot
  std::vector<int16_t> slot_exp;    // deterministic build: [sum nslots + 1] exponent bound of the slot's row sums
  std::vector<int32_t> perm;        // [rows] schedule row -> original row (empty = identity)
  std::vector<int32_t> fold_dst;    // [F] fold destination, original local row index
  std::vector<uint32_t> rowinfo;    // [nvrows] virtual row -> local_row | npackets<<16
  std::vector<V> diag;              // [nvrows] diagonal (first chunk of a row only)
  int64_t nvrows = 0;
  std::vector<SliceMeta> slice_meta; // [S]
  BigVec<V> vals;                   // [stream_len + pad] packet stream
  BigVec<uint16_t> slots;           // [slot_len + pad]: only the leader lanes' slots
  std::vector<uint8_t> leadlane;    // [S * 64 + pad] lane -> its leader lane in the slice
  int64_t slot_len = 0;
  std::vector<V> cvals;             // [coo_len] COO leftovers, packet layout
  std::vector<uint16_t> crows, ccols; // [coo_len]
  int64_t coo_len = 0, coo_entries = 0;
  // FAR sections (HYB): value, own row slot, ORIGINAL global column; packet layout
  std::vector<V> fvals;
  std::vector<uint16_t> frows;
  std::vector<int32_t> fcols;
  int64_t far_len = 0, far_entries = 0; // padded length; far nonzeros (each stored twice)
  // keep_value_map: position in the caller's values[] of every entry of vals / cvals /
  // fvals / diag (-1: padding, or a missing diagonal)
  BigVec<int32_t> val_map;
  std::vector<int32_t> cval_map, fval_map, diag_map;
  // halo fold (destinations inside [row_begin,row_end)), local row indices
  std::vector<int32_t> fold_row, fold_ptr, fold_idx;
  // remote contributions (destinations < row_begin), global row indices
  std::vector<int32_t> send_row, send_ptr, send_idx, send_counts;
  // receive side (filled by set_recv)
  int nrecv = 0;
  std::vector<int32_t> rfold_row, rfold_ptr, rfold_idx;
  int64_t stream_len = 0;
  int64_t nhalo = 0; // halo slots (halo_col carries one extra padding entry)
  bool mirrored = false;    // shard built with mirror_offblock (no sends)
  int64_t mirror_entries = 0; // one-sided entries stored for rows of higher ranks
  int64_t onesided_slots = 0; // halo slots without a y window
  int64_t far_candidates = 0; // entries the first cut marked as far (before the final cut)
  int64_t chained_packets = 0, lane_packets = 0; // packets of lanes that hand their products to a
                                                 // sibling (leadlane bit 6) / of all lanes
  std::vector<int32_t> tile_rounds; // [T] packet rounds of a tile: sum over its slices of the
                                    // longest lane's packet count (issue cost, not bytes)
  std::string error;
};

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// CFS_PLAN_VERBOSE=1 prints where tune() spends its host time
struct PhaseTimer {
  bool on;
  std::chrono::steady_clock::time_point t;
  PhaseTimer() : on(getenv("CFS_PLAN_VERBOSE") != nullptr), t(std::chrono::steady_clock::now()) {}
  void lap(const char *what) {
    if (!on) return;
    auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[cfs_plan] %-28s %8.3f s\n", what, std::chrono::duration<double>(n - t).count());
    t = n;
  }
};

// a persistent group should own at least one slice worth of rows (measured:
// small matrices are fastest with as many groups as there are resident
// workgroup slots, down to 64 rows per group)
inline int min_rows_per_group() { return 64; }

// Position of entry (lane l, diagonal j of the packet) inside a packet that
// covers `cnt` lanes (4*cnt entries).  Values: two fully contiguous
// 16-byte-per-lane loads for fp64 ([half][lane][2]), one for fp32 ([lane][4]).
// Slots: [lane][4].
template <typename V> inline int packet_val_pos(int l, int j, int cnt = 64) {
  if (sizeof(V) == 8) return (j >> 1) * 2 * cnt + l * 2 + (j & 1);
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

// The launch layout: `ngroups` cost-balanced chunks of rows, one per persistent
// workgroup -- as many as are co-resident, fewer for a small matrix.
// What a group costs beyond the bytes it streams, in byte-equivalents of one
// workgroup's share of the HBM rate.  Fitted to the per-workgroup timeline of the tile
// kernel on the Flan stand-in (tools/calib_probe.py): a workgroup's end time grows by
// 4-5 us per tile beyond its first (window epilogue + prologue, the split's own halo)
// and by 1 ns per halo slot.  The 10-20 two-tile groups of a launch -- the ragged last
// clusters of every clustering sweep, 2.4x the median halo -- used to end 10 us after
// the rest and set its length.  The model is used twice: clusters grow to equal MODEL
// cost (cluster_rows_part), and the launch order puts the expensive groups into the
// first dispatch wave (Builder::finish).  Coefficients between 3 and 8 us / 0.5 and
// 1.5 ns measure the same (+4 % on the headline); larger ones lose again.
constexpr double kHbmRate = 6.2e12;       // B/s the tile kernel sustains (section 4 of DESIGN.md)
inline double env_or(const char *name, double dflt) {
  const char *e = getenv(name);
  return e ? atof(e) : dflt;
}
static const double kTileSeconds = 1e-6 * env_or("CFS_HIP_TILE_US", 5.5);   // per tile beyond the first of a group
static const double kHaloSeconds = 1e-9 * env_or("CFS_HIP_HALO_NS", 1.0);   // per halo slot
struct ClusterCost {
  int max_slots = 0;   // LDS window of a tile (0: byte cost only)
  double per_halo = 0; // byte-equivalents per halo slot
  double per_tile = 0; // byte-equivalents per tile beyond the first
};
struct ChunkLayout {
  int block = 0, max_slots = 0, ngroups = 0, wg_per_cu = 1;
  bool full_grid = false; // as many groups as workgroups are co-resident
  int nchunks() const { return ngroups; }
  // relative cost share of every chunk, in chunk order
  std::vector<double> shares(const Options &opt) const {
    std::vector<double> s((size_t)ngroups, 1.0);
    if ((int)opt.group_share.size() == ngroups)
      for (int g = 0; g < ngroups; g++) s[g] = std::max(opt.group_share[g], 1e-9);
    return s;
  }
  ClusterCost cluster_cost(const Options &opt) const {
    ClusterCost c;
    if (!opt.cost_model) return c;
    c.max_slots = max_slots;
    c.per_halo = kHaloSeconds * kHbmRate / std::max(ngroups, 1);
    c.per_tile = kTileSeconds * kHbmRate / std::max(ngroups, 1);
    return c;
  }
};
// LDS bytes of one slot: x window in V, y window always fp64 -- two integer words and a 16-bit
// scale exponent in the deterministic build
template <typename V> inline int slot_lds_bytes(bool deterministic) {
  return (int)sizeof(V) + (deterministic ? 18 : 8);
}
template <typename V> inline ChunkLayout chunk_layout(int rows, const Options &opt) {
  ChunkLayout L;
  const int slot_bytes = slot_lds_bytes<V>(opt.deterministic);
  L.block = opt.block_threads > 0 ? opt.block_threads : kDefaultBlock;
  int max_slots = opt.max_slots > 0 ? opt.max_slots : kDefaultSlots;
  if (opt.deterministic && opt.max_slots <= 0) // still two workgroups per CU
    max_slots = std::min(max_slots, (160 * 1024 / 2 - 64) / slot_bytes / 64 * 64);
  max_slots = std::min(max_slots, (160 * 1024 - 64) / slot_bytes); // 16 B static LDS (tickets)
  max_slots = std::min(max_slots, 65536);
  max_slots = std::min(max_slots, kSlotsPerThread * std::max(L.block, 64));
  L.max_slots = std::max(max_slots, 64);
  const int64_t lds_budget = (int64_t)((L.max_slots + 63) / 64 * 64) * slot_bytes + kStaticLds;
  // co-resident workgroups per CU: LDS, the 2048-thread limit and -- the tile
  // kernel is held to 128 VGPRs -- 4 waves per SIMD; the creator passes the
  // occupancy the runtime reports for the real kernel when a device is there
  int wg_per_cu = (int)std::min<int64_t>(160 * 1024 / lds_budget, (4 * 4 * 64) / std::max(L.block, 64));
  if (opt.wg_per_cu > 0) wg_per_cu = opt.wg_per_cu;
  if (wg_per_cu < 1) wg_per_cu = 1;
  const int ncu = opt.num_cus > 0 ? opt.num_cus : 256;
  int ngroups = (ncu * wg_per_cu + 7) / 8 * 8;
  const int by_rows = ((rows + min_rows_per_group() - 1) / min_rows_per_group() + 7) / 8 * 8;
  L.full_grid = ngroups <= by_rows;
  if (ngroups > by_rows) ngroups = by_rows;
  if (ngroups < 8) ngroups = 8;
  L.ngroups = ngroups;
  L.wg_per_cu = wg_per_cu;
  return L;
}

// Which workgroup runs which group.  With two workgroups per CU the dispatcher
// places workgroups 0 .. G/2-1 first (one per CU) and G/2 .. G-1 as the second
// workgroup of the same CUs: those start 2 us later and -- the older workgroup of a
// CU wins its arbitration -- end 5 us later (tools/calib_probe.py).  The expensive
// groups of an XCD's run (several tiles, ragged clusters with much halo: the last
// clusters of every clustering sweep) therefore go FIRST, and the second workgroup
// of a CU is the cheapest partner for its first: longest-processing-time pairing
// on the model cost of a group.  Same XCD, same L2 as before.
// (cost: [rows + 1] prefix of the row costs, local row index = row - rb)
inline void compute_launch_order(const ChunkLayout &L, const Options &opt, const std::vector<Tile> &tiles,
                                 const std::vector<int32_t> &group_ptr, const std::vector<int64_t> &cost,
                                 int rb, std::vector<int32_t> &launch_order) {
  const int nc = L.nchunks();
  launch_order.resize(nc);
  for (int g = 0; g < nc; g++) launch_order[g] = g;
  if (opt.cost_model && L.full_grid && L.wg_per_cu == 2 && nc % 16 == 0 && env_or("CFS_HIP_LAUNCH_ORDER", 1) != 0) {
    const ClusterCost cm = L.cluster_cost(opt);
    std::vector<double> gc(nc, 0.0);
    for (int g = 0; g < nc; g++) {
      const int t0 = group_ptr[g], t1 = group_ptr[g + 1];
      for (int ti = t0; ti < t1; ti++) gc[g] += (double)(tiles[ti].nslots - tiles[ti].nown) * cm.per_halo;
      gc[g] += (double)std::max(0, t1 - t0 - 1) * cm.per_tile;
      if (t1 > t0) gc[g] += (double)(cost[tiles[t1 - 1].row0 + tiles[t1 - 1].nown - rb] - cost[tiles[t0].row0 - rb]);
    }
    const int nper = nc / 8, h = nper / 2;
    std::vector<int32_t> idx(nper);
    for (int x = 0; x < 8; x++) {
      for (int k = 0; k < nper; k++) idx[k] = x * nper + k;
      std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return gc[a] > gc[b]; });
      for (int j = 0; j < h; j++) {
        launch_order[x * nper + j] = idx[j];                // first wave: heaviest first
        launch_order[x * nper + h + j] = idx[nper - 1 - j]; // its CU partner: lightest
      }
    }
  }
}

// ---------------------------------------------------------------------------
// The builder: the phases of one schedule build, in the order they run
//   count_rows   stored (near) entries per row, cost prefix
//   cut_chunks   rows -> cost-balanced chunks, one per persistent workgroup
//   cut_tiles    chunks -> LDS-sized tiles (greedy, halo slot budget)
//   mark_far     HYB: halo columns a tile uses <= far_uses times -> far entries
//   resolve_far  after a cut: far entries per row, both sides
//   size_streams virtual rows, slices, leaders, stream / slot / COO / far offsets
//   fill_streams packets, slot tables, COO and FAR sections
//   fold_index   strips -> destination rows (local fold / sends of a shard)
//   finish       LDS window, per-chunk first tiles, original numbering
// ---------------------------------------------------------------------------
template <typename V> struct Builder {
  const int n;
  const int *rowptr, *colind;
  const V *values;
  const int nranks, rank;
  const Options &opt;
  const std::vector<int32_t> *chunks_in, *perm_in;
  const int32_t *srcpos; // schedule-space entry -> position in the caller's values[] (NULL: identity)
  SymPlan<V> &P;
  int rb = 0, re = 0, rows = 0;
  bool mirror = false;
  ChunkLayout L;
  int64_t max_tile_nnz = 0;
  std::vector<int32_t> lcnt, firstcol, farL, farU; // per local row
  std::vector<uint64_t> farbits;                   // one bit per CSR position (empty: none)
  std::vector<int64_t> cost;                       // [rows+1] prefix of the row costs
  std::vector<int32_t> chunk;                      // [nchunks+1] row boundaries
  std::vector<int32_t> tile_of_row;                // local row -> tile
  std::vector<int64_t> tile_len, tile_slen;
  bool mirror_fail = false, dup_error = false;
  PhaseTimer pt;

  struct VRow {
    int32_t r, k0, a; // local row, first packet of the chunk, packets in the chunk
  };
  std::vector<std::vector<VRow>> vrows; // [T] the virtual rows of every tile, slice order

  Builder(int n_, const int *rp, const int *ci, const V *va, int nranks_, int rank_,
          const Options &o, const std::vector<int32_t> *chunks, const std::vector<int32_t> *perm,
          const int32_t *src, SymPlan<V> &plan)
      : n(n_), rowptr(rp), colind(ci), values(va), nranks(nranks_), rank(rank_), opt(o),
        chunks_in(chunks), perm_in(perm), srcpos(src), P(plan) {}

  bool is_far(int64_t j) const { return !farbits.empty() && ((farbits[j >> 6] >> (j & 63)) & 1); }
  // which entries of row i does the schedule store?  The strict lower triangle,
  // and -- for a mirrored shard -- the entries right of the block (rows of higher
  // ranks that hold a_ci: their transposed update of y_i is computed HERE)
  bool stored(int i, int c) const { return c < i || (mirror && c >= re); }
  // ... in the packet / COO streams (a far entry lives in the far section)
  bool near(int i, int64_t j) const { return stored(i, colind[j]) && !is_far(j); }
  int orig(int c) const { return (perm_in && c >= rb && c < re) ? (*perm_in)[c - rb] : c; }

  // natural order: `values` is the caller's full CSR and the value of a mirrored
  // entry (i, c), c >= re, is the LOWER entry (c, i) -- all the reference's SSS
  // path reads.  Clustered order (perm_in): the caller resolved it already.
  V val_at(int i, int j) {
    const int c = colind[j];
    if (!mirror || c < re || perm_in) return values[j];
    int b = rowptr[c], e = rowptr[c + 1], l = b, r = e;
    while (l < r) {
      int m = (l + r) >> 1;
      if (colind[m] < i) l = m + 1;
      else r = m;
    }
    if (l < e && colind[l] == i) return values[l];
    for (int q = b; q < e; q++)
      if (colind[q] == i) return values[q];
#pragma omp atomic write
    mirror_fail = true; // structurally unsymmetric input
    return V(0);
  }
  // ... and where that value sits in the caller's values[] (keep_value_map)
  int32_t src_at(int i, int j) const {
    if (srcpos) return srcpos[j];
    const int c = colind[j];
    if (!mirror || c < re) return j;
    int b = rowptr[c], e = rowptr[c + 1], l = b, r = e;
    while (l < r) {
      int m = (l + r) >> 1;
      if (colind[m] < i) l = m + 1;
      else r = m;
    }
    if (l < e && colind[l] == i) return l;
    for (int q = b; q < e; q++)
      if (colind[q] == i) return q;
    return -1;
  }

  bool setup(const int *row_splits_in) {
    P = SymPlan<V>();
    P.n = n;
    P.nranks = nranks;
    P.rank = rank;
    if (row_splits_in) P.row_splits.assign(row_splits_in, row_splits_in + nranks + 1);
    else P.row_splits = {0, n};
    rb = P.row_begin = P.row_splits[rank];
    re = P.row_end = P.row_splits[rank + 1];
    if (rb < 0 || re > n || rb > re) {
      P.error = "bad row_splits";
      return false;
    }
    rows = re - rb;
    const int block = opt.block_threads > 0 ? opt.block_threads : kDefaultBlock;
    if (block != 256 && block != 512 && block != 1024) {
      P.error = "block_threads must be 256, 512 or 1024";
      return false;
    }
    L = chunk_layout<V>(rows, opt);
    if (chunks_in) { // the caller's clusters: same layout rule, boundaries given
      if ((int)chunks_in->size() != L.nchunks() + 1) {
        P.error = "internal: chunk count does not match the launch layout";
        return false;
      }
    }
    P.block_threads = L.block;
    P.wg_per_cu = L.wg_per_cu;
    P.max_slots = L.max_slots;
    P.ngroups = L.ngroups;
    P.deterministic = opt.deterministic;
    max_tile_nnz = opt.max_tile_nnz > 0 ? opt.max_tile_nnz : (int64_t)1 << 30;
    mirror = opt.mirror_offblock && nranks > 1;
    P.mirrored = mirror;
    return true;
  }

  // ---- stored entries per row (the far ones apart), cost prefix --------------------
  bool count_rows() {
    lcnt.assign(rows, 0);
    firstcol.assign(rows, -1);
    int64_t nnz_low = 0, nnz_diag = 0, nnz_mirror = 0, mirror_dup = 0;
#pragma omp parallel for schedule(static) reduction(+ : nnz_low, nnz_diag, nnz_mirror, mirror_dup) num_threads(host_threads())
    for (int i = rb; i < re; i++) {
      int c = 0, up = 0, prev_up = -1, nearc = 0, fc = -1;
      bool dup = false;
      for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
        const int col = colind[j];
        if (col < i) c++;
        else if (col == i) nnz_diag++;
        else if (mirror && col >= re) {
          if (col <= prev_up) dup = true; // duplicate / unsorted: cannot pair with (c, i)
          prev_up = col;
          up++;
        }
        if (near(i, j)) {
          nearc++;
          // first stored column: rows of one mesh node share it (sibling test)
          if (fc < 0) fc = col;
        }
      }
      lcnt[i - rb] = nearc;
      firstcol[i - rb] = fc;
      nnz_low += c;
      nnz_mirror += up;
      if (dup) mirror_dup++;
    }
    if (mirror_dup) {
      P.error = "mirror: duplicate or unsorted off-block entries";
      return false;
    }
    P.mirror_entries = nnz_mirror;
    P.nnz_low = nnz_low;
    P.nnz_diag = nnz_diag;
    P.nnz_full = 2 * nnz_low + nnz_diag;
    cost.assign((size_t)rows + 1, 0);
    const bool far = !farL.empty();
    for (int r = 0; r < rows; r++)
      cost[r + 1] = cost[r] + (int64_t)lcnt[r] * (int64_t)(sizeof(V) + 2) +
                    (far ? (int64_t)(farL[r] + farU[r]) * (int64_t)(2 * sizeof(V) + 6) : 0) +
                    (int64_t)(4 + 3 * sizeof(V)) + 2 * (int64_t)sizeof(V);
    return true;
  }

  // ---- rows -> chunks of equal streamed bytes ------------------------------------------
  // A CU only streams ~1/256 of the HBM bandwidth, so the launch is as long as
  // its most loaded CU: rows are first cut into contiguous chunks of equal streamed
  // bytes (one chunk per persistent workgroup, as many workgroups as are
  // co-resident), and only then into LDS-sized tiles that never straddle a chunk.
  void cut_chunks() {
    const int nc = L.nchunks();
    if (chunks_in) {
      chunk = *chunks_in;
      return;
    }
    chunk.assign((size_t)nc + 1, re);
    chunk[0] = rb;
    const std::vector<double> share = L.shares(opt);
    std::vector<double> cum((size_t)nc + 1, 0.0);
    for (int c = 0; c < nc; c++) cum[c + 1] = cum[c] + share[c];
    for (int c = 1; c < nc; c++) {
      const int64_t target = (int64_t)((double)cost[rows] * (cum[c] / cum[nc]));
      int r = (int)(std::lower_bound(cost.begin(), cost.end(), target) - cost.begin());
      if (r > rows) r = rows;
      if (rb + r < chunk[c - 1]) r = chunk[c - 1] - rb;
      chunk[c] = rb + r;
    }
    chunk[nc] = re;
  }

  // ---- chunks -> tiles: greedy cut under the slot budget and a cost cap ---------------
  // Chunks are independent: one thread each, with its own "column already counted for
  // tile id" stamps.  A far entry takes no slot.
  bool cut_tiles(std::vector<Tile> &tiles, std::vector<int32_t> &group_ptr) {
    const int nc = L.nchunks(), max_slots = L.max_slots;
    std::vector<std::vector<Tile>> per_chunk(nc);
    std::string cut_error;
#pragma omp parallel num_threads(host_threads())
    {
      std::vector<int32_t> stamp(n > 0 ? n : 1, -1);
      int tid = 0;
      std::string err;
      auto cut = [&](int r0, int r1, int64_t cap, std::vector<Tile> &out) -> bool {
        int row = r0;
        while (row < r1) {
          Tile t{};
          t.row0 = row;
          int nown = 0, nhalo = 0;
          int64_t nnz = 0, c0 = cost[row - rb];
          while (row < r1) {
            int newh = 0, len = 0;
            for (int j = rowptr[row]; j < rowptr[row + 1]; j++) {
              const int c = colind[j];
              if (!stored(row, c)) continue;
              len++;
              if (is_far(j)) continue;
              if ((c < t.row0 || c >= re) && stamp[c] != tid) {
                stamp[c] = tid;
                newh++;
              }
            }
            if (len > 65535) {
              err = "row with more than 65535 lower entries";
              return false;
            }
            const bool fits = (nown + 1 + nhalo + newh <= max_slots) &&
                              (nnz + len <= max_tile_nnz || nown == 0) && nown < 65535 &&
                              (cost[row + 1 - rb] - c0 <= cap || nown == 0);
            if (!fits) {
              if (nown == 0) {
                err = "a single row needs more LDS slots than max_slots "
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
          out.push_back(t);
          tid++;
        }
        return true;
      };
#pragma omp for schedule(dynamic, 1)
      for (int g = 0; g < nc; g++) {
        const int r0 = chunk[g], r1 = chunk[g + 1];
        if (r0 >= r1 || !err.empty()) continue;
        std::vector<Tile> &tmp = per_chunk[g];
        if (!cut(r0, r1, (int64_t)1 << 60, tmp)) continue;
        if (tmp.size() > 1) { // even the tiles of a chunk out (less halo, same count)
          const int64_t cc = cost[r1 - rb] - cost[r0 - rb];
          std::vector<Tile> even;
          if (!cut(r0, r1, cc / (int64_t)tmp.size() + cc / 64 + 1, even)) continue;
          if (even.size() <= tmp.size()) tmp.swap(even);
        }
      }
      if (!err.empty()) {
#pragma omp critical
        cut_error = err;
      }
    }
    if (!cut_error.empty()) {
      P.error = cut_error;
      return false;
    }
    tiles.clear();
    group_ptr.assign((size_t)nc + 1, 0);
    for (int g = 0; g < nc; g++) {
      group_ptr[g] = (int32_t)tiles.size();
      tiles.insert(tiles.end(), per_chunk[g].begin(), per_chunk[g].end());
    }
    group_ptr[nc] = (int32_t)tiles.size();
    return true;
  }

  // ---- HYB, first half: which entries leave the tile format? -----------------------------
  // For the tiles of a first cut: an in-block halo column that its tile uses at most
  // far_uses times.  (Off-block columns of a shard keep their slots: they are
  // one-sided already, or their sums are sent to the owner.)
  void mark_far(const std::vector<Tile> &tiles) {
    const int64_t nnz = rowptr[n];
    farbits.assign((size_t)(nnz + 63) / 64 + 1, 0);
    const int T = (int)tiles.size(), thr = std::max(1, opt.far_uses);
    int64_t marked = 0;
#pragma omp parallel num_threads(host_threads()) reduction(+ : marked)
    {
      std::vector<int32_t> uses(n > 0 ? n : 1, 0);
      std::vector<int32_t> touched;
#pragma omp for schedule(dynamic, 4)
      for (int ti = 0; ti < T; ti++) {
        const Tile &t = tiles[ti];
        touched.clear();
        for (int i = t.row0; i < t.row0 + t.nown; i++)
          for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
            const int c = colind[j];
            if (c < t.row0 && c >= rb && c < i) {
              if (uses[c]++ == 0) touched.push_back(c);
            }
          }
        for (int i = t.row0; i < t.row0 + t.nown; i++)
          for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
            const int c = colind[j];
            if (c < t.row0 && c >= rb && c < i && uses[c] <= thr) {
              __atomic_fetch_or(&farbits[(size_t)j >> 6], 1ull << (j & 63), __ATOMIC_RELAXED);
              marked++;
            }
          }
        for (int c : touched) uses[c] = 0;
      }
    }
    P.far_candidates = marked;
    if (marked == 0) farbits.clear();
  }

  // ---- HYB, second half: after a cut ------------------------------------------------------
  // A marked entry whose two rows ended up in one tile is an ordinary entry again.
  // farL / farU = far entries of a row as the lower / the mirrored (upper) end.
  void resolve_far(const std::vector<Tile> &tiles) {
    tile_of_row.assign(rows, 0);
    const int T = (int)tiles.size();
#pragma omp parallel for schedule(static) num_threads(host_threads())
    for (int ti = 0; ti < T; ti++)
      for (int r = 0; r < tiles[ti].nown; r++) tile_of_row[tiles[ti].row0 - rb + r] = ti;
    farL.assign(rows, 0);
    farU.assign(rows, 0);
    P.far_entries = 0;
    if (farbits.empty()) return;
    int64_t kept = 0;
#pragma omp parallel for schedule(static) reduction(+ : kept) num_threads(host_threads())
    for (int i = rb; i < re; i++)
      for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
        if (!is_far(j)) continue;
        const int c = colind[j];
        if (tile_of_row[c - rb] == tile_of_row[i - rb]) {
          __atomic_fetch_and(&farbits[(size_t)j >> 6], ~(1ull << (j & 63)), __ATOMIC_RELAXED);
          continue;
        }
        farL[i - rb]++;
        __atomic_fetch_add(&farU[c - rb], 1, __ATOMIC_RELAXED);
        kept++;
      }
    P.far_entries = kept;
    if (kept == 0) farbits.clear();
  }

  // ---- virtual rows: one lane each ----------------------------------------------
  // A lane normally owns a whole row.  A row much longer than its neighbours
  // (irregular degrees, e.g. ldoor) would leave one lane streaming alone for
  // most of its slice, so such a row is split into chunks of at most `acap`
  // packets; every chunk is a VIRTUAL ROW with its own lane, the chunks add
  // their partial sums to the same LDS slot and only the first carries the
  // diagonal.  Virtual rows are sorted by packet count (stable counting sort,
  // descending): the lanes still active in packet g of a slice are a prefix.
  void build_vrows(const Tile &t, std::vector<VRow> &vr) const {
    vr.clear();
    int64_t sum = 0;
    for (int r = 0; r < t.nown; r++) sum += lcnt[t.row0 - rb + r] >> 2;
    const int avg = (int)((sum + t.nown - 1) / std::max(1, (int)t.nown));
    const int acap = std::max(8, 2 * avg);
    // SIBLINGS stay together: consecutive rows with the same first stored column
    // (the rows of one mesh node: their column sequences are prefixes of one
    // another) form a group; groups are sorted by their longest member
    // (stable counting sort, descending), slices are cut from that order and
    // only then sorted by packet count inside each slice.  A sibling with fewer
    // packets can then name a longer one IN ITS SLICE as its slot leader.
    struct Grp {
      int32_t first, count, key; // range in tmp, max packets of a member
    };
    std::vector<VRow> tmp;
    std::vector<Grp> grp;
    int maxa = 0;
    for (int r = 0; r < t.nown; r++) {
      const int a = lcnt[t.row0 - rb + r] >> 2;
      const bool split = a > acap;
      const bool join = !grp.empty() && !split && grp.back().count < 16 && grp.back().key <= acap &&
                        firstcol[t.row0 - rb + r] >= 0 &&
                        firstcol[t.row0 - rb + r] == firstcol[t.row0 - rb + r - 1];
      if (!join) grp.push_back(Grp{(int32_t)tmp.size(), 0, 0});
      if (!split) {
        tmp.push_back(VRow{r, 0, a});
        grp.back().count++;
        grp.back().key = std::max(grp.back().key, (int32_t)a);
        maxa = std::max(maxa, a);
      } else {
        const int parts = (a + acap - 1) / acap;
        for (int q = 0, k0 = 0; q < parts; q++) {
          const int ca = (a - k0 + (parts - q) - 1) / (parts - q); // even chunks
          tmp.push_back(VRow{r, k0, ca});
          grp.back().count++;
          grp.back().key = std::max(grp.back().key, (int32_t)(acap + 1)); // never joined
          maxa = std::max(maxa, ca);
          k0 += ca;
        }
      }
    }
    // sort key of a split row's group: its longest chunk
    for (auto &g : grp)
      if (g.key > acap) {
        int k = 0;
        for (int q = 0; q < g.count; q++) k = std::max(k, (int)tmp[g.first + q].a);
        g.key = k;
      }
    std::vector<int32_t> cnt(maxa + 2, 0);
    for (auto &g : grp) cnt[maxa - g.key + 1] += g.count;
    for (int k = 0; k <= maxa; k++) cnt[k + 1] += cnt[k];
    vr.resize(tmp.size());
    for (auto &g : grp) {
      int32_t &pos = cnt[maxa - g.key];
      for (int q = 0; q < g.count; q++) vr[pos++] = tmp[g.first + q];
    }
    for (size_t p0 = 0; p0 < vr.size(); p0 += kLanes) { // inside a slice: longest first
      const size_t p1 = std::min(vr.size(), p0 + (size_t)kLanes);
      std::stable_sort(vr.begin() + p0, vr.begin() + p1,
                       [](const VRow &x, const VRow &y) { return x.a > y.a; });
    }
  }

  // `cnt` near columns of row i from its k0-th on, stored order
  void near_cols(int i, int k0, int cnt, std::vector<int32_t> &out) const {
    out.clear();
    int seen = 0;
    for (int j = rowptr[i]; j < rowptr[i + 1] && (int)out.size() < cnt; j++)
      if (near(i, j)) {
        if (seen >= k0) out.push_back(colind[j]);
        seen++;
      }
  }

  // ---- per tile: virtual rows, slices, leaders; then all offsets ---------------------------
  // Multi-dof FEM matrices repeat themselves: the rows of one mesh node have
  // (nearly) the same columns.  A lane whose packet-covered column sequence is
  // a PREFIX of the sequence of an earlier lane of its slice (earlier = at
  // least as many packets) does not store slots at all: it reads the slots of
  // that LEADER lane (same addresses -> the same cache lines serve the run).
  // Per slice: a 64-bit leader mask, and one byte per lane naming its leader.
  bool size_streams() {
    const int T = (int)P.tiles.size();
    tile_len.assign(T, 0);
    tile_slen.assign(T, 0);
    vrows.assign(T, std::vector<VRow>());
#pragma omp parallel num_threads(host_threads())
    {
#pragma omp for schedule(dynamic, 1)
      for (int ti = 0; ti < T; ti++) {
        Tile &t = P.tiles[ti];
        std::vector<VRow> &vr = vrows[ti];
        build_vrows(t, vr);
        t.nvrows = (int32_t)vr.size();
        t.nslices = (t.nvrows + kLanes - 1) / kLanes;
        int64_t left = 0, far = 0, farlow = 0;
        for (int r = 0; r < t.nown; r++) {
          left += lcnt[t.row0 - rb + r] & 3;
          if (!farbits.empty()) {
            farlow += farL[t.row0 - rb + r];
            far += farL[t.row0 - rb + r] + farU[t.row0 - rb + r];
          }
        }
        t.ncoo = (int32_t)left;
        t.nfar = (int32_t)far;
        t.nfar_low = (int32_t)farlow;
      }
    }
    int64_t halo = 0, slices = 0, nsl = 0, nvr = 0, far = 0;
    for (auto &t : P.tiles) {
      t.halo_off = (int32_t)halo;
      t.slice_base = (int32_t)slices;
      t.slot_off = (int32_t)nsl;
      t.vrow_off = (int32_t)nvr;
      t.far_off = (int32_t)far;
      halo += t.nslots - t.nown;
      nsl += t.nslots;
      slices += t.nslices;
      nvr += t.nvrows;
      far += align_up(t.nfar, 256);
      if (halo > 0x7fffffffLL || nsl > 0x7ffffff0LL || nvr > 0x7ffffff0LL || far > 0x7fffff00LL) {
        P.error = "halo index overflow";
        return false;
      }
    }
    P.nhalo = halo;
    P.far_len = far;
    P.halo_col.assign((size_t)halo + 1, 0);
    P.slot_col.assign((size_t)nsl + 1, 0); // +1: the kernel's clamped dummy read
    P.slice_meta.assign((size_t)slices, SliceMeta{0u, 0u, 0ull});
    P.leadlane.assign((size_t)slices * kLanes + kLanes, 0);
    P.rowinfo.assign((size_t)nvr + 1, 0);
    P.diag.assign((size_t)nvr + 1, V(0));
    P.nvrows = nvr;
#pragma omp parallel num_threads(host_threads())
    {
      std::vector<int32_t> cur;
      std::vector<std::vector<int32_t>> lseq(kLanes); // sequences of the slice's leaders
#pragma omp for schedule(dynamic, 1)
      for (int ti = 0; ti < T; ti++) {
        Tile &t = P.tiles[ti];
        const std::vector<VRow> &vr = vrows[ti];
        int64_t off = 0, soff = 0;
        for (int s = 0; s < t.nslices; s++) {
          off = align_up(off, kAlignEntries);
          soff = align_up(soff, 4);
          int p0 = s * kLanes, p1 = std::min(p0 + kLanes, (int)t.nvrows);
          uint32_t cnt0 = 0;
          uint64_t leaders = 0;
          SliceMeta &sm = P.slice_meta[t.slice_base + s];
          sm.voff = (uint32_t)off;
          if (soff >= (1 << 25)) tile_slen[ti] = -1; // flagged below
          sm.soff_cnt0 = (uint32_t)soff;
          uint8_t *ll = P.leadlane.data() + (size_t)(t.slice_base + s) * kLanes;
          for (int p = p0; p < p1; p++) {
            const VRow &v = vr[p];
            const int l = p - p0;
            off += (int64_t)v.a * 4;
            if (v.a >= 1) cnt0++;
            int lead = l;
            if (v.a >= 1) {
              near_cols(t.row0 + v.r, v.k0 * 4, v.a * 4, cur);
              // latest leaders first: siblings sit next to each other
              for (int j = l - 1; j >= 0 && lead == l; j--)
                if (((leaders >> j) & 1) && lseq[j].size() >= cur.size() && !lseq[j].empty() &&
                    lseq[j][0] == cur[0] && std::equal(cur.begin(), cur.end(), lseq[j].begin()))
                  lead = j;
            }
            ll[l] = (uint8_t)lead;
            if (lead == l) {
              leaders |= 1ull << l;
              soff += (int64_t)v.a * 4;
              if (v.a >= 1) lseq[l].swap(cur);
              else lseq[l].clear();
            }
          }
          for (int l = p1 - p0; l < kLanes; l++) { // unused lanes: own (empty) runs
            leaders |= 1ull << l;
            ll[l] = (uint8_t)l;
          }
          // Lanes that read the same slots add to the same y-window words.  A follower
          // that sits right behind a lane of its group (its leader or another follower),
          // inside the same row of 16 lanes, hands its products to that lane instead (DPP
          // row shift in the kernel, runs of at most three lanes): bit 6 = "my products go
          // to my left neighbour", bit 7 = "I take my right neighbour's".
          if (opt.combine_siblings) {
            int run = 0; // lanes already chained to the current head
            bool cf[kLanes];
            for (int l = 0; l < kLanes; l++) {
              const bool chained = l > 0 && (l & 15) != 0 && (ll[l] & 63) != l && l < p1 - p0 &&
                                   (ll[l] & 63) == (ll[l - 1] & 63) && run < 2;
              cf[l] = chained;
              run = chained ? run + 1 : 0;
            }
            int64_t chained = 0, all = 0;
            for (int l = 0; l < kLanes; l++) {
              ll[l] = (uint8_t)((ll[l] & 63) | (cf[l] ? 64 : 0) | (l + 1 < kLanes && cf[l + 1] ? 128 : 0));
              if (l < p1 - p0) {
                all += vr[p0 + l].a;
                if (cf[l]) chained += vr[p0 + l].a;
              }
            }
            __atomic_fetch_add(&P.chained_packets, chained, __ATOMIC_RELAXED);
            __atomic_fetch_add(&P.lane_packets, all, __ATOMIC_RELAXED);
          }
          sm.soff_cnt0 |= cnt0 << 25;
          sm.leaders = leaders;
        }
        tile_len[ti] = align_up(off, kAlignEntries);
        if (tile_slen[ti] >= 0) tile_slen[ti] = align_up(soff, kAlignEntries);
      }
    }
    int64_t off = 0, soff = 0, coo = 0;
    for (int ti = 0; ti < T; ti++) {
      if (tile_slen[ti] < 0) {
        P.error = "tile slot stream exceeds 2^25 entries";
        return false;
      }
      P.tiles[ti].nnz_off = off;
      off += tile_len[ti];
      P.tiles[ti].sl_off = soff;
      soff += tile_slen[ti];
      P.tiles[ti].coo_off = (int32_t)coo;
      P.coo_entries += P.tiles[ti].ncoo;
      coo += align_up(P.tiles[ti].ncoo, 256);
      if (coo > 0x7fffff00LL) {
        P.error = "COO section overflow";
        return false;
      }
    }
    P.stream_len = off;
    P.slot_len = soff;
    P.coo_len = coo;
    // one packet of padding: the kernel prefetches a slice's first packet with
    // every lane before it knows how many lanes the packet really has
    par_assign(P.vals, (size_t)off + kStreamPad, V(0)); // zeros: the padding of the packets
    par_assign(P.slots, (size_t)soff + kStreamPad, (uint16_t)0);
    P.cvals.assign((size_t)coo + 256, V(0));
    P.crows.assign((size_t)coo + 256, 0);
    P.ccols.assign((size_t)coo + 256, 0);
    P.fvals.assign((size_t)P.far_len + 256, V(0));
    P.frows.assign((size_t)P.far_len + 256, 0);
    P.fcols.assign((size_t)P.far_len + 256, 0);
    if (opt.keep_value_map) {
      par_assign(P.val_map, (size_t)off + kStreamPad, (int32_t)-1);
      P.cval_map.assign((size_t)coo + 256, -1);
      P.fval_map.assign((size_t)P.far_len + 256, -1);
      P.diag_map.assign((size_t)nvr + 1, -1);
    }
    return true;
  }

  // ---- fill: halo slot tables, packet streams, COO leftovers, far sections -------------
  bool fill_streams() {
    const int T = (int)P.tiles.size();
    P.tile_rounds.assign(T, 0);
    struct FarE {
      int32_t r, c; // own local row, ORIGINAL global column
      V v;
      int32_t src; // position of the value in the caller's values[]
    };
    const bool keep = opt.keep_value_map;
    // the mirrored (upper) ends of the far entries are found from the lower side:
    // a cursor per tile, filled in parallel, sorted afterwards for a fixed order
    std::vector<std::vector<FarE>> far_up(farbits.empty() ? 0 : T);
    if (!farbits.empty()) {
      std::vector<int32_t> cur(T, 0);
      for (int ti = 0; ti < T; ti++)
        far_up[ti].resize((size_t)(P.tiles[ti].nfar - P.tiles[ti].nfar_low));
#pragma omp parallel for schedule(dynamic, 256) num_threads(host_threads())
      for (int i = rb; i < re; i++)
        for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
          if (!is_far(j)) continue;
          const int c = colind[j], tc = tile_of_row[c - rb];
          const int at = __atomic_fetch_add(&cur[tc], 1, __ATOMIC_RELAXED);
          far_up[tc][at] = FarE{c - P.tiles[tc].row0, orig(i), values[j], keep ? src_at(i, j) : -1};
        }
    }
#pragma omp parallel num_threads(host_threads())
    {
      std::vector<int32_t> colmap(n > 0 ? n : 1, -1); // col -> halo slot (per thread)
      std::vector<int32_t> hcols, lowj;
      std::vector<FarE> fl;
#pragma omp for schedule(dynamic, 1)
      for (int ti = 0; ti < T; ti++) {
        const Tile &t = P.tiles[ti];
        // halo: unique near columns outside the tile; in-block columns (they get
        // a y window entry and a strip entry) first, then the off-block columns of
        // a mirrored shard (x only), each class ascending
        hcols.clear();
        for (int r = 0; r < t.nown; r++) {
          const int i = t.row0 + r;
          for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
            const int c = colind[j];
            if (near(i, j) && (c < t.row0 || c >= re) && colmap[c] < 0) {
              colmap[c] = 0;
              hcols.push_back(c);
            }
          }
        }
        auto offblock = [&](int c) { return mirror && (c < rb || c >= re); };
        std::sort(hcols.begin(), hcols.end(), [&](int a, int b) {
          const bool oa = offblock(a), ob = offblock(b);
          return oa != ob ? ob : a < b;
        });
        bool bad = (int)hcols.size() != t.nslots - t.nown;
        {
          int ny = t.nown;
          for (int c : hcols)
            if (!offblock(c)) ny++;
          P.tiles[ti].ny = ny;
        }
        for (size_t h = 0; h < hcols.size(); h++) {
          colmap[hcols[h]] = t.nown + (int)h;
          if (!bad) P.halo_col[t.halo_off + h] = hcols[h];
        }
        auto slot_of = [&](int c) {
          return (uint16_t)((c >= t.row0 && c < re) ? c - t.row0 : colmap[c]);
        };
        // positions (in the CSR) of the near entries of local row r, in stored
        // order -- works whether or not the columns of a row ascend
        auto lower_of = [&](int r, std::vector<int32_t> &out) {
          out.clear();
          const int i = t.row0 + r;
          for (int j = rowptr[i]; j < rowptr[i + 1]; j++)
            if (near(i, j)) out.push_back(j);
        };
        const std::vector<VRow> &vr = vrows[ti];
        V *tv = P.vals.data() + t.nnz_off;
        uint16_t *ts = P.slots.data() + t.sl_off;
        std::vector<std::vector<int32_t>> low(kLanes);
        double amax_t = 0.0; // largest |value| of the tile
        int nonfinite_t = 0;  // ... and whether a NaN / Inf is among its values
        for (int s = 0; s < t.nslices && !bad; s++) {
          int p0 = s * kLanes, m = std::min(kLanes, (int)t.nvrows - p0);
          int amax = 0;
          for (int l = 0; l < m; l++) {
            const VRow &v = vr[p0 + l];
            const int i = t.row0 + v.r;
            lower_of(v.r, low[l]);
            amax = std::max(amax, v.a);
            P.rowinfo[t.vrow_off + p0 + l] = (uint32_t)v.r | ((uint32_t)v.a << 16);
            V d = V(0);
            int dj = -1;
            if (v.k0 == 0)
              for (int j = rowptr[i]; j < rowptr[i + 1]; j++)
                if (colind[j] == i) {
                  d = values[j]; // last duplicate wins, like :1292
                  dj = j;
                }
            P.diag[t.vrow_off + p0 + l] = d;
            if (keep && dj >= 0) P.diag_map[t.vrow_off + p0 + l] = srcpos ? srcpos[dj] : dj;
            amax_t = std::max(amax_t, std::fabs((double)d));
            nonfinite_t |= !std::isfinite((double)d);
          }
          const SliceMeta &sm = P.slice_meta[t.slice_base + s];
          int64_t o = sm.voff, os = sm.soff_cnt0 & 0x1ffffff;
          P.tile_rounds[ti] += amax;
          for (int g = 0; g < amax; g++) {
            int cnt = 0;
            while (cnt < m && vr[p0 + cnt].a > g) cnt++;
            int lead = -1; // index of lane l's leader among the leaders of this packet
            for (int l = 0; l < cnt; l++) {
              const bool is_leader = (sm.leaders >> l) & 1;
              if (is_leader) lead++;
              for (int j = 0; j < kPacket; j++) {
                int q = low[l][(vr[p0 + l].k0 + g) * kPacket + j];
                const V av = val_at(t.row0 + vr[p0 + l].r, q);
                tv[o + packet_val_pos<V>(l, j, cnt)] = av;
                if (keep)
                  P.val_map[t.nnz_off + o + packet_val_pos<V>(l, j, cnt)] = src_at(t.row0 + vr[p0 + l].r, q);
                amax_t = std::max(amax_t, std::fabs((double)av));
                nonfinite_t |= !std::isfinite((double)av);
                if (is_leader) ts[os + packet_slot_pos(lead, j)] = slot_of(colind[q]);
              }
            }
            o += 4 * (int64_t)cnt;
            os += 4 * (int64_t)(lead + 1);
          }
        }
        // COO leftovers: the last len%4 near entries of every row, natural row
        // order, in 256-entry packets with the packet value layout
        if (!bad) {
          V *cv = P.cvals.data() + t.coo_off;
          uint16_t *cr = P.crows.data() + t.coo_off, *cc = P.ccols.data() + t.coo_off;
          int64_t e = 0;
          for (int r = 0; r < t.nown; r++) {
            lower_of(r, lowj);
            int len = (int)lowj.size();
            for (int k = (len >> 2) << 2; k < len && e < t.ncoo; k++) {
              int q = lowj[k];
              int64_t pk = e >> 8;
              int l = (int)((e & 255) >> 2), j = (int)(e & 3);
              const V av = val_at(t.row0 + r, q);
              cv[pk * 256 + packet_val_pos<V>(l, j)] = av;
              if (keep) P.cval_map[t.coo_off + pk * 256 + packet_val_pos<V>(l, j)] = src_at(t.row0 + r, q);
              amax_t = std::max(amax_t, std::fabs((double)av));
              nonfinite_t |= !std::isfinite((double)av);
              cr[pk * 256 + packet_slot_pos(l, j)] = (uint16_t)r;
              cc[pk * 256 + packet_slot_pos(l, j)] = slot_of(colind[q]);
              e++;
            }
          }
          if (e != t.ncoo) bad = true;
        }
        // FAR section: own rows' far entries in row order, then the mirror images of
        // other tiles' far entries that end in this tile's rows (sorted: fixed order)
        if (t.nfar > 0 && !bad) {
          fl.clear();
          for (int r = 0; r < t.nown; r++) {
            const int i = t.row0 + r;
            for (int j = rowptr[i]; j < rowptr[i + 1]; j++)
              if (is_far(j)) fl.push_back(FarE{r, orig(colind[j]), values[j], keep ? src_at(i, j) : -1});
          }
          if ((int)fl.size() != t.nfar_low) bad = true;
          std::vector<FarE> &up = far_up[ti];
          std::sort(up.begin(), up.end(), [](const FarE &a, const FarE &b) {
            return a.r != b.r ? a.r < b.r : a.c < b.c;
          });
          fl.insert(fl.end(), up.begin(), up.end());
          if ((int)fl.size() != t.nfar) bad = true;
          V *fv = P.fvals.data() + t.far_off;
          uint16_t *fr = P.frows.data() + t.far_off;
          int32_t *fc = P.fcols.data() + t.far_off;
          for (size_t e = 0; e < fl.size() && (int64_t)e < t.nfar; e++) {
            const int64_t pk = (int64_t)e >> 8;
            const int l = (int)((e & 255) >> 2), j = (int)(e & 3);
            fv[pk * 256 + packet_val_pos<V>(l, j)] = fl[e].v;
            if (keep) P.fval_map[t.far_off + pk * 256 + packet_val_pos<V>(l, j)] = fl[e].src;
            amax_t = std::max(amax_t, std::fabs((double)fl[e].v));
            nonfinite_t |= !std::isfinite((double)fl[e].v);
            fr[pk * 256 + packet_slot_pos(l, j)] = (uint16_t)fl[e].r;
            fc[pk * 256 + packet_slot_pos(l, j)] = fl[e].c;
          }
        }
        // (std::max drops a NaN operand: a NaN value is caught by its own test)
        P.tiles[ti].aexp = !(nonfinite_t == 0) ? kAexpNonFinite
                           : (amax_t > 0.0 && std::isfinite(amax_t)) ? std::ilogb(amax_t) + 1 : -1000;
        if (bad) {
#pragma omp atomic write
          dup_error = true;
        }
        for (int c : hcols) colmap[c] = -1;
      }
    }
    if (dup_error) {
      P.error = "internal: halo / leftover / far count mismatch";
      return false;
    }
    if (mirror_fail) {
      P.error = "mirror: structurally unsymmetric off-block entries";
      return false;
    }
    if (getenv("CFS_PLAN_VERBOSE"))
      fprintf(stderr, "[cfs_plan] slot stream %lld entries for %lld value entries; far entries %lld (stored twice)\n",
              (long long)P.slot_len, (long long)P.stream_len, (long long)P.far_entries);
    return true;
  }

  // ---- halo fold index: strips -> destination rows, fixed order --------------
  void fold_index() {
    const int64_t H = P.nhalo;
    std::vector<int32_t> lcount(rows + 1, 0), rcount(rb + 1, 0);
    for (int64_t q = 0; q < H; q++) {
      int c = P.halo_col[q];
      if (c >= rb && c < re) lcount[c - rb + 1]++;
      else if (mirror) P.onesided_slots++; // x only: nothing to fold, nothing to send
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
      if (c >= rb && c < re) P.fold_idx[lfill[lpos[c - rb]]++] = (int32_t)q;
      else if (!mirror) P.send_idx[rfill[rpos[c]]++] = (int32_t)q;
    }
    P.send_counts.assign(nranks, 0);
    for (int r : P.send_row) {
      int owner = (int)(std::upper_bound(P.row_splits.begin(), P.row_splits.end(), r) -
                        P.row_splits.begin()) - 1;
      P.send_counts[owner]++;
    }
    if (getenv("CFS_PLAN_VERBOSE")) {
      int64_t mx = 0, long8 = 0, sum_long = 0;
      for (size_t i = 0; i + 1 < P.fold_ptr.size(); i++) {
        int64_t l = P.fold_ptr[i + 1] - P.fold_ptr[i];
        mx = std::max(mx, l);
        if (l > 8) {
          long8++;
          sum_long += l - 8;
        }
      }
      fprintf(stderr, "[cfs_plan] fold: %zu destinations, longest list %lld, %lld lists > 8 (%lld entries beyond)\n",
              P.fold_row.size(), (long long)mx, (long long)long8, (long long)sum_long);
    }
  }

  // ---- LDS window, per-chunk first tiles, schedule space -> original indices -------------
  void finish() {
    int lds_slots = 64;
    for (auto &t : P.tiles) lds_slots = std::max(lds_slots, (int)t.nslots);
    P.lds_slots = (lds_slots + 63) / 64 * 64;
    const int nc = L.nchunks();
    P.group_first.assign(nc, Tile{});
    for (int g = 0; g < nc; g++)
      if (P.group_ptr[g] < P.group_ptr[g + 1]) P.group_first[g] = P.tiles[P.group_ptr[g]];
    // the kernels address x and y in the caller's (original) numbering
    for (const Tile &t : P.tiles) {
      for (int i = 0; i < t.nown; i++) P.slot_col[t.slot_off + i] = orig(t.row0 + i);
      for (int h = 0; h < t.nslots - t.nown; h++)
        P.slot_col[t.slot_off + t.nown + h] = orig(P.halo_col[t.halo_off + h]);
    }
    if (opt.deterministic && opt.row_exp) { // scale of every slot's fixed-point sums
      P.slot_exp.assign(P.slot_col.size(), 0);
      for (size_t q = 0; q + 1 < P.slot_col.size(); q++) P.slot_exp[q] = opt.row_exp[P.slot_col[q]];
    }
    P.fold_dst.resize(P.fold_row.size());
    for (size_t i = 0; i < P.fold_row.size(); i++) P.fold_dst[i] = orig(P.fold_row[i] + rb) - rb;
    if (perm_in) P.perm = *perm_in;
    compute_launch_order(L, opt, P.tiles, P.group_ptr, cost, rb, P.launch_order);
  }

  // the whole build; cut_only: the caller only wants to compare halo sizes of two row orders
  bool run(const int *row_splits_in, bool cut_only) {
    if (!setup(row_splits_in)) return false;
    if (!count_rows()) return false;
    pt.lap("core: row counts");
    cut_chunks();
    if (opt.hyb && rows > 0) {
      std::vector<Tile> t1;
      std::vector<int32_t> gp1;
      if (!cut_tiles(t1, gp1)) return false;
      mark_far(t1);
      pt.lap("core: far marks");
    }
    if (!cut_tiles(P.tiles, P.group_ptr)) return false;
    if (!opt.hyb && opt.count_far && rows > 0 && !cut_only) { // what would HYB take out? (tune())
      mark_far(P.tiles);
      farbits.clear();
    }
    resolve_far(P.tiles);
    if (!farbits.empty()) {
      // the far entries are known now: near counts, costs and -- in natural order -- the
      // chunk boundaries once more, then the final tiles.  A marked entry stays far
      // exactly when its rows are in different tiles of THE FINAL cut (entries that
      // became local are near again, which can only lower the slot demand of a tile).
      if (!count_rows()) return false;
      cut_chunks();
      if (!cut_tiles(P.tiles, P.group_ptr)) return false;
      resolve_far(P.tiles);
      if (!count_rows()) return false;
    }
    pt.lap("core: cut tiles");
    if (cut_only) {
      P.nhalo = 0;
      for (const Tile &t : P.tiles) P.nhalo += t.nslots - t.nown;
      return true;
    }
    if (!size_streams()) return false;
    pt.lap("core: vrows + sizes");
    if (!fill_streams()) return false;
    pt.lap("core: fill streams");
    fold_index();
    pt.lap("core: fold index");
    finish();
    return true;
  }
};

// Build the plan for rows [row_splits[rank], row_splits[rank+1]) of the CSR
// (natural order: the caller's full CSR; clustered: its lower triangle in schedule
// space).  Returns false (plan.error set) when the matrix cannot be scheduled.
template <typename V>
bool build_plan_core(int n, const int *rowptr, const int *colind, const V *values,
                     int nranks, int rank, const int *row_splits_in,
                     const Options &opt, const std::vector<int32_t> *chunks_in,
                     const std::vector<int32_t> *perm_in, SymPlan<V> &P, bool cut_only = false,
                     const int32_t *srcpos = nullptr) {
  Builder<V> b(n, rowptr, colind, values, nranks, rank, opt, chunks_in, perm_in, srcpos, P);
  return b.run(row_splits_in, cut_only);
}

// Greedy graph growing: order the rows [rb, re) so that `ngroups` consecutive
// chunks of (almost exactly) equal streamed cost are compact clusters of the
// matrix graph -- the role METIS/KaHIP play for the reference's
// partition_by_conflicts (csr_matrix.tpp:543-639): fewer columns shared
// between partitions = fewer halo slots here, fewer conflicts there.  A
// cluster grows breadth-first from a seed over unassigned rows until its cost
// share is reached; what is left of its frontier seeds the next clusters, so
// the clusters sweep through the graph like a wavefront.  O(nnz).
// The cost of a cluster is exact: a row's stored (lower) entries in the new
// order are its neighbours left of the block plus its neighbours assigned
// before it, and the sum over a cluster does not depend on the order inside.
// one sweep over the rows [pb, pe) of the block [.., block_re): columns left of pb are
// outside (their entries are stored at the row whatever the order), columns in
// [pe, block_re) belong to a later sweep (numbered later: not stored at this row)
template <typename V>
void cluster_rows_part(const int *rowptr, const int *colind, int pb, int pe, int block_re,
                       int ngroups, const double *share, std::vector<int32_t> &perm,
                       std::vector<int32_t> &chunk, bool mirror, int64_t total_known = -1,
                       const ClusterCost &cm = ClusterCost(), int ncols = 0) {
  const int rows = pe - pb;
  const int64_t per_nz = (int64_t)sizeof(V) + 2, per_row = 4 + 5 * (int64_t)sizeof(V);
  int64_t total = total_known;
  if (total < 0) { // (the caller of the parts has counted already)
    int64_t left = 0, inblock = 0;
#pragma omp parallel for schedule(static) reduction(+ : left, inblock) num_threads(host_threads()) if (!omp_in_parallel())
    for (int i = pb; i < pe; i++)
      for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
        int c = colind[j];
        if (c < pb || (mirror && c >= block_re)) left++;
        else if (c < pe && c != i) inblock++;
      }
    total = (int64_t)rows * per_row + (left + inblock / 2) * per_nz;
  }
  perm.clear();
  perm.reserve(rows);
  chunk.assign(ngroups + 1, pe);
  chunk[0] = pb;
  std::vector<int32_t> state(rows, -1); // -1 free, -2-g queued for cluster g, >=0 assigned
  std::vector<int32_t> bfs, seeds;
  size_t seed_head = 0;
  int next_free = 0;
  std::vector<double> cumshare(ngroups + 1, 0.0);
  for (int g = 0; g < ngroups; g++) cumshare[g + 1] = cumshare[g] + std::max(share[g], 1e-9);
  // the cost model (ClusterCost): a cluster also pays for its halo slots -- distinct
  // columns outside it that its rows store: rows of earlier clusters, columns left of
  // the part -- and for every LDS window beyond its first.  What that adds up to is
  // only known at the end, so every cluster's target is its share of what is LEFT:
  // the bytes not yet assigned plus the extras the remaining clusters are expected
  // to pay (at the rate seen so far).
  const bool model = cm.max_slots > 0 && ncols > 0;
  struct Zeroed { // lazily zeroed pages: only the marks that are touched cost memory
    int32_t *p;
    explicit Zeroed(size_t k) : p(k ? (int32_t *)calloc(k, sizeof(int32_t)) : nullptr) {}
    ~Zeroed() { free(p); }
  } inmark(model ? (size_t)rows : 0), outmark(model ? (size_t)ncols : 0);
  int64_t cum_bytes = 0;
  double cum_extra = 0.0;
  for (int g = 0; g < ngroups; g++) {
    const double s_rem = cumshare[ngroups] - cumshare[g];
    const double extra_rate = g > 0 ? cum_extra / cumshare[g] : 0.0;
    const double target = ((double)(total - cum_bytes) + extra_rate * s_rem) * (cumshare[g + 1] - cumshare[g]) / s_rem;
    const bool last = g == ngroups - 1;
    const size_t first = perm.size();
    bfs.clear();
    size_t head = 0;
    int64_t bytes_g = 0, halo_g = 0, rows_g = 0;
    double extra_g = 0.0;
    while ((int)perm.size() < rows && (last || (double)bytes_g + extra_g < target)) {
      if (head == bfs.size()) { // need a seed: oldest frontier row, else next free row
        int sd = -1;
        while (seed_head < seeds.size()) {
          int v = seeds[seed_head++];
          if (state[v] < 0) {
            sd = v;
            break;
          }
        }
        if (sd < 0) {
          while (next_free < rows && state[next_free] >= 0) next_free++;
          if (next_free >= rows) break;
          sd = next_free;
        }
        state[sd] = -2 - g;
        bfs.push_back(sd);
      }
      const int v = bfs[head++];
      int low = 0;
      const int i = pb + v;
      for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
        const int c = colind[j];
        if (c < pb || (mirror && c >= block_re)) {
          low++;
          if (model && (unsigned)c < (unsigned)ncols && outmark.p[c] != g + 1) {
            outmark.p[c] = g + 1;
            halo_g++;
          }
        } else if (c < pe && c != i) {
          const int u = c - pb;
          if (state[u] >= 0) {
            low++;
            if (model && state[u] != g && inmark.p[u] != g + 1) {
              inmark.p[u] = g + 1;
              halo_g++;
            }
          } else if (state[u] != -2 - g) {
            state[u] = -2 - g;
            bfs.push_back(u);
          }
        }
      }
      state[v] = g;
      perm.push_back(i);
      rows_g++;
      bytes_g += (int64_t)low * per_nz + per_row;
      if (model)
        extra_g = (double)halo_g * cm.per_halo +
                  (double)((rows_g + halo_g - 1) / cm.max_slots) * cm.per_tile;
    }
    cum_bytes += bytes_g;
    cum_extra += extra_g;
    for (size_t k = head; k < bfs.size(); k++) { // unfinished frontier -> future seeds
      state[bfs[k]] = -1;
      seeds.push_back(bfs[k]);
    }
    std::sort(perm.begin() + first, perm.end()); // original order inside a cluster
    chunk[g + 1] = pb + (int)perm.size();
  }
}

// The sweep is sequential by nature; a large block is cut into up to 8 PARTS of
// consecutive rows (equal shares of the cost, whole clusters each) that are swept
// independently, one host thread each.  Across a part boundary the natural order
// decides which end stores an entry, as it does across the tiles of the natural
// order; inside a part nothing changes.  (A small block keeps the single sweep: that
// is what numbers the hub rows of an arrow matrix early.)
inline int kMaxParts() {
  if (const char *e = getenv("CFS_CLUSTER_PARTS")) return std::max(1, atoi(e));
  return 8;
}
template <typename V>
void cluster_rows(int n, const int *rowptr, const int *colind, int rb, int re, int ngroups,
                  const std::vector<double> &share_in, std::vector<int32_t> &perm,
                  std::vector<int32_t> &chunk, bool mirror = false,
                  const ClusterCost &cm = ClusterCost()) {
  const int rows = re - rb;
  std::vector<double> share(ngroups, 1.0);
  if ((int)share_in.size() == ngroups) share = share_in;
  int parts = 1;
  if (rows >= 200000 && ngroups >= 64)
    while (parts * 2 <= std::min(host_threads(), kMaxParts()) && ngroups % (parts * 2) == 0 &&
           ngroups / (parts * 2) >= 16)
      parts *= 2;
  if (parts == 1) {
    cluster_rows_part<V>(rowptr, colind, rb, re, re, ngroups, share.data(), perm, chunk, mirror, -1, cm, n);
    return;
  }
  // part boundaries: the cost every order agrees on (an entry across a boundary is
  // stored at its higher row either way), split by the shares of the parts' clusters
  const int64_t per_nz = (int64_t)sizeof(V) + 2, per_row = 4 + 5 * (int64_t)sizeof(V);
  std::vector<int64_t> cost((size_t)rows + 1, 0);
#pragma omp parallel for schedule(static) num_threads(host_threads())
  for (int i = rb; i < re; i++) {
    int low = 0;
    for (int j = rowptr[i]; j < rowptr[i + 1]; j++)
      if (colind[j] < i || (mirror && colind[j] >= re)) low++;
    cost[i - rb + 1] = (int64_t)low * per_nz + per_row;
  }
  for (int r = 0; r < rows; r++) cost[r + 1] += cost[r];
  const int gpp = ngroups / parts;
  std::vector<double> cum(parts + 1, 0.0);
  for (int p = 0; p < parts; p++) {
    cum[p + 1] = cum[p];
    for (int g = 0; g < gpp; g++) cum[p + 1] += std::max(share[p * gpp + g], 1e-9);
  }
  std::vector<int> pb(parts + 1, re);
  pb[0] = rb;
  for (int p = 1; p < parts; p++) {
    const int64_t target = (int64_t)((double)cost[rows] * (cum[p] / cum[parts]));
    int r = (int)(std::lower_bound(cost.begin(), cost.end(), target) - cost.begin());
    pb[p] = std::max(pb[p - 1], rb + std::min(r, rows));
  }
  std::vector<int64_t> part_left(parts, 0), part_in(parts, 0);
  { // parts only pay when the given order has locality: most neighbours of a row must
    // lie in its own part (a randomly numbered mesh has them everywhere -- one sweep then)
    // (the same pass counts what every part's sweep needs: its total cost)
    int64_t inpart = 0, inblock = 0;
#pragma omp parallel num_threads(host_threads())
    {
      std::vector<int64_t> l_left(parts, 0), l_in(parts, 0);
#pragma omp for schedule(static) reduction(+ : inpart, inblock) nowait
      for (int i = rb; i < re; i++) {
        const int p = (int)(std::upper_bound(pb.begin(), pb.end(), i) - pb.begin()) - 1;
        for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
          const int c = colind[j];
          if (c < pb[p] || (mirror && c >= re)) l_left[p]++;
          else if (c < pb[p + 1] && c != i) l_in[p]++;
          if (c < rb || c >= re || c == i) continue;
          inblock++;
          if (c >= pb[p] && c < pb[p + 1]) inpart++;
        }
      }
      for (int p = 0; p < parts; p++) {
#pragma omp atomic
        part_left[p] += l_left[p];
#pragma omp atomic
        part_in[p] += l_in[p];
      }
    }
    if (inpart * 10 < inblock * 8) {
      cluster_rows_part<V>(rowptr, colind, rb, re, re, ngroups, share.data(), perm, chunk, mirror, -1, cm, n);
      return;
    }
  }
  std::vector<std::vector<int32_t>> pperm(parts), pchunk(parts);
#pragma omp parallel for schedule(static, 1) num_threads(parts)
  for (int p = 0; p < parts; p++)
    cluster_rows_part<V>(rowptr, colind, pb[p], pb[p + 1], re, gpp, share.data() + p * gpp,
                         pperm[p], pchunk[p], mirror,
                         (int64_t)(pb[p + 1] - pb[p]) * per_row + (part_left[p] + part_in[p] / 2) * per_nz,
                         cm, n);
  perm.clear();
  perm.reserve(rows);
  chunk.assign(ngroups + 1, re);
  for (int p = 0; p < parts; p++) {
    perm.insert(perm.end(), pperm[p].begin(), pperm[p].end());
    for (int g = 0; g <= gpp; g++) chunk[p * gpp + g] = pchunk[p][g]; // row counts == positions
  }
  chunk[ngroups] = re;
}

// The clustered row order of a block and its matrix in schedule space: what a
// second build of the same rows with HALF as many groups can reuse (tune() tries
// two window shapes: the clusters of the coarser schedule are pairs of the finer
// one's, consecutive in the sweep order).
template <typename V> struct ScheduleSpace {
  bool valid = false;
  bool device_only = false; // perm / chunk only: the matrix arrays live on the GPU (cfs_devplan.hpp)
  std::shared_ptr<void> device_keep; // ... the uploaded matrix and its clustered placement (cfs_dev::Kept)
  int rb = 0, re = 0, nchunks = 0;
  std::vector<int32_t> perm, chunk, brp;
  BigVec<int32_t> bci;
  BigVec<V> bva;
  BigVec<int32_t> bsr; // keep_value_map: entry -> position in the caller's values[]
  ScheduleSpace() = default;
  ScheduleSpace(const ScheduleSpace &) = delete;
  ScheduleSpace &operator=(const ScheduleSpace &) = delete;
  ~ScheduleSpace() { drop(); }
  void drop() {
    valid = false;
    device_only = false;
    device_keep.reset();
    release_async(bci);
    release_async(bva);
    release_async(bsr);
    std::vector<int32_t>().swap(brp);
  }
};

// deterministic build: 2^e > 1-norm of every row of the caller's matrix (both triangles)
template <typename V>
void compute_row_exp(int n, const int *rowptr, const V *values, std::vector<int16_t> &row_exp) {
  row_exp.assign((size_t)n, (int16_t)-1000);
#pragma omp parallel for schedule(static) num_threads(host_threads())
  for (int i = 0; i < n; i++) {
    double sum = 0.0;
    for (int j = rowptr[i]; j < rowptr[i + 1]; j++) sum += std::fabs((double)values[j]);
    row_exp[i] = !std::isfinite(sum) ? kExpNonFinite : (sum > 0.0 ? (int16_t)(std::ilogb(sum) + 1) : (int16_t)-1000);
  }
}

// Build the plan for rows [row_splits[rank], row_splits[rank+1]) of the full
// CSR.  Returns false (plan.error set) when the matrix cannot be scheduled.
template <typename V>
bool build_plan(int n, const int *rowptr, const int *colind, const V *values, int nranks,
                int rank, const int *row_splits_in, const Options &opt_in, SymPlan<V> &P,
                ScheduleSpace<V> *cache = nullptr) {
  Options opt = opt_in;
  std::vector<int16_t> row_exp;
  if (opt.deterministic && n > 0 && rowptr && values) {
    compute_row_exp<V>(n, rowptr, values, row_exp);
    opt.row_exp = row_exp.data();
  }
  const int rb = row_splits_in ? row_splits_in[rank] : 0;
  const int re = row_splits_in ? row_splits_in[rank + 1] : n;
  const int rows = re - rb;
  if (opt.mirror_offblock && nranks > 1 && rb >= 0 && re <= n && rb <= re) {
    // A mirrored shard finds the entries (r, c) of higher ranks' rows r through
    // their mirror images (c, r) in its OWN rows.  Every such image is matched
    // to its lower entry when the streams are filled; equal counts then make the
    // match a bijection.  A structurally unsymmetric matrix is refused here
    // (the exchange form reads the lower triangle only and takes it).
    int64_t up = 0, low_in = 0;
#pragma omp parallel for schedule(static) reduction(+ : up) num_threads(host_threads())
    for (int i = rb; i < re; i++)
      for (int j = rowptr[i]; j < rowptr[i + 1]; j++)
        if (colind[j] >= re) up++;
#pragma omp parallel for schedule(static) reduction(+ : low_in) num_threads(host_threads())
    for (int r = re; r < n; r++)
      for (int j = rowptr[r]; j < rowptr[r + 1]; j++)
        if (colind[j] >= rb && colind[j] < re) low_in++;
    if (up != low_in) {
      P = SymPlan<V>();
      P.error = "mirror: structurally unsymmetric off-block entries";
      return false;
    }
  }
  const int block = opt.block_threads > 0 ? opt.block_threads : kDefaultBlock;
  if (!opt.reorder || opt.force_order == 1 || rb < 0 || re > n || rows < 256 ||
      (block != 256 && block != 512 && block != 1024)) // (the core reports a bad block)
    return build_plan_core<V>(n, rowptr, colind, values, nranks, rank, row_splits_in, opt,
                              nullptr, nullptr, P);
  // the launch layout: same rule as the core
  const ChunkLayout L = chunk_layout<V>(rows, opt);
  const int nchunks = L.nchunks();

  PhaseTimer pt;
  const bool mirror = opt.mirror_offblock && nranks > 1;
  ScheduleSpace<V> local;
  ScheduleSpace<V> &sp = cache ? *cache : local;
  if (cache && cache->valid && !cache->device_only && cache->rb == rb && cache->re == re &&
      cache->nchunks == 2 * nchunks && opt.group_share.empty()) {
    // reuse: same row order, every second cluster boundary; always clustered
    std::vector<int32_t> merged(nchunks + 1);
    for (int g = 0; g <= nchunks; g++) merged[g] = sp.chunk[2 * g];
    pt.lap("schedule space reused");
    if (build_plan_core<V>(n, sp.brp.data(), sp.bci.data(), sp.bva.data(), nranks, rank,
                           row_splits_in, opt, &merged, &sp.perm, P, false,
                           sp.bsr.empty() ? nullptr : sp.bsr.data()))
      return true;
    return build_plan_core<V>(n, rowptr, colind, values, nranks, rank, row_splits_in, opt,
                              nullptr, nullptr, P);
  }
  sp.valid = false;
  sp.device_only = false;
  std::vector<int32_t> &perm = sp.perm, &chunk = sp.chunk;
  cluster_rows<V>(n, rowptr, colind, rb, re, nchunks, L.shares(opt), perm, chunk, mirror,
                  L.cluster_cost(opt));
  pt.lap("cluster_rows");
  std::vector<int32_t> inv(rows);
  for (int p = 0; p < rows; p++) inv[perm[p] - rb] = rb + p;
  // the lower triangle + diagonal in schedule space.  Every stored LOWER entry
  // (i, c), c <= i, of the block -- all the reference's SSS path reads -- lands on
  // exactly one schedule row: the later one of the two rows it couples, with the
  // earlier one as its column (off-block columns keep their original numbering and
  // stay with row i).  A scatter, not a gather: no entry has to look for its mirror
  // image.  A mirrored shard also keeps the UPPER off-block entries (i, c >= re) of
  // its rows, one-sided, valued from the lower entry (c, i) of the rank above.
  auto lower_value_pos = [&](int hi, int lo) -> int { // position of (hi, lo), hi > lo
    int b = rowptr[hi], e = rowptr[hi + 1];
    int l = b, r = e;
    while (l < r) { // columns usually ascend: binary search first
      int m = (l + r) >> 1;
      if (colind[m] < lo) l = m + 1;
      else r = m;
    }
    if (l < e && colind[l] == lo) return l;
    for (int j = b; j < e; j++)
      if (colind[j] == lo) return j; // unsorted row
    return -1;
  };
  std::vector<int32_t> &brp = sp.brp;
  brp.assign((size_t)n + 2, 0);
  {
    int32_t *cnt = brp.data() + 1; // cnt[p]: entries of schedule row p
#pragma omp parallel for schedule(static) num_threads(host_threads())
    for (int i = rb; i < re; i++) {
      const int p = inv[i - rb];
      int own = 0;
      for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
        const int c = colind[j];
        if (c > i) {
          own += mirror && c >= re;
        } else if (c == i || c < rb) {
          own++;
        } else {
          const int pc = inv[c - rb];
          if (pc < p) own++;
          else __atomic_fetch_add(&cnt[pc], 1, __ATOMIC_RELAXED);
        }
      }
      __atomic_fetch_add(&cnt[p], own, __ATOMIC_RELAXED);
    }
  }
  for (int p = rb; p < re; p++) brp[p + 1] += brp[p]; // rows < rb are empty
  for (int p = re + 1; p <= n + 1; p++) brp[p] = brp[re]; // rows >= re too (brp[n] = nnz)
  const int64_t bnnz = brp[re];
  // (not filled: the scatter below writes every entry, its threads touch the pages)
  BigVec<int32_t> &bci = sp.bci;
  BigVec<V> &bva = sp.bva;
  bci.clear();
  bva.clear();
  bci.resize((size_t)bnnz + 1);
  bva.resize((size_t)bnnz + 1);
  bci[bnnz] = 0;
  bva[bnnz] = V(0);
  BigVec<int32_t> &bsr = sp.bsr;
  bsr.clear();
  if (opt.keep_value_map) {
    bsr.resize((size_t)bnnz + 1);
    bsr[bnnz] = -1;
  }
  bool asym = false;
  {
    std::vector<int32_t> cur(brp.begin() + rb, brp.begin() + re); // fill cursor per schedule row
    const bool maps = !bsr.empty();
#pragma omp parallel for schedule(static) num_threads(host_threads())
    for (int i = rb; i < re; i++) {
      const int p = inv[i - rb];
      for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
        const int c = colind[j];
        int row = p, col, src = j;
        if (c > i) {
          if (!(mirror && c >= re)) continue;
          col = c;
          src = lower_value_pos(c, i);
          if (src < 0) {
#pragma omp atomic write
            asym = true; // no lower image: the mirror check refuses such blocks anyway
            src = j;
          }
        } else if (c == i) {
          col = p;
        } else if (c < rb) {
          col = c;
        } else {
          const int pc = inv[c - rb];
          if (pc < p) col = pc;
          else row = pc, col = p;
        }
        const int q = __atomic_fetch_add(&cur[row - rb], 1, __ATOMIC_RELAXED);
        bci[q] = col;
        bva[q] = values[src];
        if (maps) bsr[q] = src;
      }
    }
  }
  // the entries of a row arrived in any order: sort them by column.  Duplicate
  // entries (the reader keeps them) would make that order -- and with it the
  // schedule -- depend on the race above: such matrices keep their natural order.
#pragma omp parallel num_threads(host_threads())
  {
    struct Ent {
      int32_t first;
      V second;
      int32_t src;
    };
    std::vector<Ent> tmp;
    const bool maps = !bsr.empty();
    bool dup = false;
#pragma omp for schedule(dynamic, 1024) nowait
    for (int p = rb; p < re; p++) {
      const int b = brp[p], e = brp[p + 1];
      bool sorted = true;
      for (int q = b + 1; q < e; q++)
        if (bci[q] <= bci[q - 1]) {
          sorted = false;
          break;
        }
      if (sorted) continue;
      tmp.resize((size_t)(e - b));
      for (int q = b; q < e; q++) tmp[q - b] = Ent{bci[q], bva[q], maps ? bsr[q] : 0};
      std::sort(tmp.begin(), tmp.end(), [](const Ent &x, const Ent &y) { return x.first < y.first; });
      for (int q = b; q < e; q++) {
        if (q > b && tmp[q - b].first == tmp[q - b - 1].first) dup = true;
        bci[q] = tmp[q - b].first;
        bva[q] = tmp[q - b].second;
        if (maps) bsr[q] = tmp[q - b].src;
      }
    }
    if (dup) {
#pragma omp atomic write
      asym = true;
    }
  }
  if (asym)
    return build_plan_core<V>(n, rowptr, colind, values, nranks, rank, row_splits_in, opt,
                              nullptr, nullptr, P);
  pt.lap("schedule-space matrix");
  // keep the clustered schedule only if it has at least 2.5x fewer halo slots:
  // clustered tiles pay for their compactness with scattered x / y accesses
  // (measured on MI355X, fp64: Flan stand-in 3.5x fewer halo slots -> 8 % faster;
  // pwtk stand-in 2.1x fewer -> 11 % slower; random sparsity has no locality to
  // find).  In fp32 the scattered 4-byte accesses weigh twice as much against a
  // stream half as long: the same 3.5x is 7 % slower, so the bar is 6x there.
  // Only the tile cut of both orders is needed to decide (with HYB: the halo that is
  // left once the far entries are gone).
  bool use_clustered = true;
  if (opt.force_order != 2) {
    SymPlan<V> C, N;
    const bool c_ok = build_plan_core<V>(n, brp.data(), bci.data(), bva.data(), nranks, rank,
                                         row_splits_in, opt, &chunk, &perm, C, true);
    const bool n_ok = build_plan_core<V>(n, rowptr, colind, values, nranks, rank, row_splits_in,
                                         opt, nullptr, nullptr, N, true);
    if (getenv("CFS_PLAN_VERBOSE"))
      fprintf(stderr, "[cfs_plan] halo slots: clustered %lld (%zu tiles), natural %lld (%zu tiles)\n",
              c_ok ? (long long)C.nhalo : -1LL, C.tiles.size(), n_ok ? (long long)N.nhalo : -1LL,
              N.tiles.size());
    if (!c_ok) use_clustered = false;
    else if (n_ok && (sizeof(V) == 8 ? 2 * N.nhalo <= 5 * C.nhalo : N.nhalo <= 6 * C.nhalo))
      use_clustered = false;
    // ... and only if the natural order is LDS-bound into more tiles than the
    // clustered one.  When a group's rows fit one window either way, clustering
    // still cuts the halo ~3x but loses (measured, Flan stand-in fp64: 1/8 shards
    // 25.1 vs 23.4 us per SpMV, 1/2 shards 66.6 vs 64.9 us) -- while the whole
    // matrix, 2 natural tiles per group against 1 clustered, gains 7 %.
    else if (n_ok && N.tiles.size() <= C.tiles.size())
      use_clustered = false;
  }
  if (use_clustered &&
      build_plan_core<V>(n, brp.data(), bci.data(), bva.data(), nranks, rank, row_splits_in, opt,
                         &chunk, &perm, P, false, bsr.empty() ? nullptr : bsr.data())) {
    sp.valid = true; // a later build with half as many chunks may reuse it
    sp.rb = rb;
    sp.re = re;
    sp.nchunks = nchunks;
    return true;
  }
  release_async(bci);
  release_async(bva);
  release_async(bsr);
  return build_plan_core<V>(n, rowptr, colind, values, nranks, rank, row_splits_in, opt, nullptr,
                            nullptr, P);
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
// lower triangle (original numbering), walking it exactly as the kernel does (packets,
// jagged diagonals, slot -> column through the own range / halo map, far sections).
// Structure check for the CPU test-suite; performs no SpMV arithmetic.  `up_*`
// (optional) receive the mirrored far entries as (higher row, lower row, value): as
// a multiset they must equal the far entries of the lower side, `far_*`.
template <typename V>
void decode_plan(const SymPlan<V> &P, std::vector<int32_t> &row,
                 std::vector<int32_t> &col, std::vector<V> &val,
                 std::vector<int32_t> *far_row = nullptr, std::vector<int32_t> *far_col = nullptr,
                 std::vector<V> *far_val = nullptr, std::vector<int32_t> *up_row = nullptr,
                 std::vector<int32_t> *up_col = nullptr, std::vector<V> *up_val = nullptr) {
  row.clear();
  col.clear();
  val.clear();
  const int rb = P.row_begin;
  auto orig = [&](int c) { return (!P.perm.empty() && c >= rb && c < P.row_end) ? P.perm[c - rb] : c; };
  for (const Tile &t : P.tiles) {
    const V *tv = P.vals.data() + t.nnz_off;
    const uint16_t *ts = P.slots.data() + t.sl_off;
    auto slot_col = [&](int s) {
      return s < t.nown ? t.row0 + s : P.halo_col[t.halo_off + (s - t.nown)];
    };
    // per row: packet entries first (k = 0 .. 4a-1), then its COO leftovers, so
    // that the decoded order inside a row equals the stored order
    std::vector<std::vector<std::pair<int32_t, V>>> rows_out(t.nown);
    for (int s = 0; s < t.nslices; s++) {
      int p0 = s * kLanes, m = std::min(kLanes, (int)t.nvrows - p0);
      int a[kLanes], r[kLanes];
      for (int l = 0; l < kLanes; l++) {
        uint32_t info = l < m ? P.rowinfo[t.vrow_off + p0 + l] : 0;
        r[l] = info & 0xffff;
        a[l] = info >> 16;
      }
      const SliceMeta &sm = P.slice_meta[t.slice_base + s];
      int64_t o = sm.voff, os = sm.soff_cnt0 & 0x1ffffff;
      int amax = a[0];
      for (int g = 0; g < amax; g++) {
        int cnt = 0;
        while (cnt < kLanes && a[cnt] > g) cnt++;
        // lane l reads the slots of its leader lane L = leadlane[l]; L's block is
        // the (number of leaders below L)-th of the packet -- every leader below L
        // has at least as many packets as L, hence is active whenever L is
        const uint8_t *ll = P.leadlane.data() + (size_t)(t.slice_base + s) * kLanes;
        int nlead = 0;
        for (int l = 0; l < cnt; l++) {
          if ((sm.leaders >> l) & 1) nlead++;
          const int Ld = ll[l] & 63;
          const int rank = __builtin_popcountll(sm.leaders & ((1ull << Ld) - 1));
          for (int j = 0; j < kPacket; j++)
            rows_out[r[l]].push_back({slot_col(ts[os + packet_slot_pos(rank, j)]),
                                      tv[o + packet_val_pos<V>(l, j, cnt)]});
        }
        o += 4 * (int64_t)cnt;
        os += 4 * (int64_t)nlead;
      }
    }
    const V *cv = P.cvals.data() + t.coo_off;
    const uint16_t *cr = P.crows.data() + t.coo_off, *cc = P.ccols.data() + t.coo_off;
    for (int64_t e = 0; e < t.ncoo; e++) {
      int64_t pk = e >> 8;
      int l = (int)((e & 255) >> 2), j = (int)(e & 3);
      rows_out[cr[pk * 256 + packet_slot_pos(l, j)]].push_back(
          {slot_col(cc[pk * 256 + packet_slot_pos(l, j)]),
           cv[pk * 256 + packet_val_pos<V>(l, j)]});
    }
    for (int rr = 0; rr < t.nown; rr++)
      for (auto &e : rows_out[rr]) {
        const int a = orig(t.row0 + rr), b = orig(e.first);
        row.push_back(std::max(a, b)); // back to the caller's lower orientation
        col.push_back(std::min(a, b));
        val.push_back(e.second);
      }
    // far section: columns are stored in ORIGINAL numbering already
    const V *fv = P.fvals.data() + t.far_off;
    const uint16_t *fr = P.frows.data() + t.far_off;
    const int32_t *fc = P.fcols.data() + t.far_off;
    for (int64_t e = 0; e < t.nfar; e++) {
      const int64_t pk = e >> 8;
      const int l = (int)((e & 255) >> 2), j = (int)(e & 3);
      const int a = orig(t.row0 + fr[pk * 256 + packet_slot_pos(l, j)]);
      const int b = fc[pk * 256 + packet_slot_pos(l, j)];
      const V v = fv[pk * 256 + packet_val_pos<V>(l, j)];
      if (e < t.nfar_low) {
        row.push_back(std::max(a, b));
        col.push_back(std::min(a, b));
        val.push_back(v);
        if (far_row) {
          far_row->push_back(std::max(a, b));
          far_col->push_back(std::min(a, b));
          far_val->push_back(v);
        }
      } else if (up_row) {
        up_row->push_back(std::max(a, b));
        up_col->push_back(std::min(a, b));
        up_val->push_back(v);
      }
    }
  }
}

} // namespace cfs_plan
