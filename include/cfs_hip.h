/*
 * cfs_hip.h -- C ABI of libcfs_hip.so: the MI355X (gfx950) symmetric-SpMV hot
 * path of cfs-spmv behind plain pointers and sizes.
 *
 * This is the drop-in boundary.  The reference (athelaf/cfs-spmv) has no FFI of
 * its own -- it is a C++ library whose hot path hides behind two seams:
 *     tune()                   include/matrix/csr_matrix.tpp:230-310
 *     dense_vector_multiply()  include/matrix/csr_matrix.hpp:67-70  (spmv_fn)
 * and the allocation seam internal_alloc/internal_free
 *     include/utils/allocator.hpp:11-12, src/allocator.cpp:8-43.
 * Each entry point below names the reference interface it replaces.  The C++
 * surface (include/cfs.hpp: SparseMatrix / CSRMatrix / SpDMV) and the Python
 * mirror (cfs_spmv_amd/) are thin callers of this ABI; INTEGRATION.md shows
 * the binding a maintainer of the reference would add.
 *
 * Conventions: every function returns 0 on success or a negative error code;
 * cfs_hip_last_error() returns the message of the calling thread's last
 * failure.  No C++ objects or exceptions cross this line.  Indices are int32
 * (only <int,float> and <int,double> are instantiated in the reference:
 * src/csr.cpp:10-11, src/cfs.cpp:11-21).
 */
#ifndef CFS_HIP_H
#define CFS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFS_HIP_ABI_VERSION 4

/* error codes */
#define CFS_HIP_OK 0
#define CFS_HIP_ERR_ARG (-1)      /* bad argument                            */
#define CFS_HIP_ERR_DEVICE (-2)   /* HIP runtime failure (message has detail) */
/* the tile schedule does not cover this matrix (a row denser than an LDS window,
 * offsets exhausted): the caller may fall back to the general CSR kernel.      */
#define CFS_HIP_ERR_UNSUPPORTED (-3)
#define CFS_HIP_ERR_NOMEM (-4)
#define CFS_HIP_ERR_INTERNAL (-5) /* a consistency check of the schedule builder failed */
/* a shard cannot be built in the mirrored form (off-block structure unsymmetric
 * or duplicated): rebuild it with CFS_HIP_FLAG_SHARD_EXCHANGE                   */
#define CFS_HIP_ERR_MIRROR (-6)

typedef struct cfs_hip_sym_s *cfs_hip_sym_t; /* symmetric (SSS) matrix handle */
typedef struct cfs_hip_csr_s *cfs_hip_csr_t; /* general CSR matrix handle     */

/* ---- runtime (replaces src/runtime.cpp:10-34: CFS_NUM_THREADS / affinity) -- */
int cfs_hip_abi_version(void);
const char *cfs_hip_last_error(void);
int cfs_hip_device_count(int *count);
/* Make `device` the calling thread's current device and the HOME of the
 * synchronous entry points (cfs_hip_alloc, host-pointer SpMV).  Idempotent, and it
 * leaves the contexts of other devices alone: handles remember the device they
 * were created on and keep working.  A process that never calls it adopts the
 * device that is current in the calling thread at the first use (never a
 * hard-wired device 0).                                                        */
int cfs_hip_init(int device);
int cfs_hip_current_device(int *device); /* the home device */
/* 1 once a home device is bound; never initialises the runtime itself (the allocator seam asks
 * before it hands out page-locked host memory: a plain host allocation must not start HIP) */
int cfs_hip_runtime_bound(void);
/* stream used internally by the synchronous (host-pointer capable) entry
 * points; created by cfs_hip_init.  The *_async entry points take the caller's
 * hipStream_t verbatim: NULL there means HIP's null stream (which is what
 * torch.cuda.current_stream() is by default), never this one.               */
int cfs_hip_default_stream(void **stream);
int cfs_hip_synchronize(void *stream); /* NULL = the library stream */

/* ---- allocator (replaces internal_alloc / internal_free,
 *      src/allocator.cpp:8-43; Platform::gpu memory)                      ---- */
#define CFS_HIP_MEM_DEVICE 0 /* hipMalloc: HBM, the residency the timed loop needs */
#define CFS_HIP_MEM_PINNED 1 /* hipHostMalloc: page-locked host staging buffer      */
int cfs_hip_alloc(size_t bytes, int kind, void **out);
int cfs_hip_free(void *p, int kind);
/* The page-locked blocks come from a pool (power-of-two classes >= 64 KiB; a
 * released block is kept for the next request).  owns: 1 if p is a live block of
 * the pool (the allocator seam frees those through cfs_hip_free).              */
int cfs_hip_pinned_owns(const void *p);
int cfs_hip_pinned_pool_stats(size_t *live_blocks, size_t *spare_blocks, size_t *spare_bytes);
#define CFS_HIP_H2D 0
#define CFS_HIP_D2H 1
#define CFS_HIP_D2D 2
int cfs_hip_memcpy(void *dst, const void *src, size_t bytes, int dir);
int cfs_hip_memset(void *dst, int value, size_t bytes);

/* ---- tuning knobs of the tile schedule (all 0 = library default) ---------- */
typedef struct {
  int max_slots;     /* LDS slots (x and y windows) per tile               */
  int max_tile_nnz;  /* cap on stored nonzeros per tile                    */
  int block_threads; /* workgroup size of the tile kernel: 256, 512, 1024  */
  int flags;         /* CFS_HIP_FLAG_*                                      */
} cfs_hip_options;
/* timing-only ablations of the tile kernel (WRONG results by construction;
 * used by tools/ to price LDS atomics and LDS gathers against the pure matrix
 * stream; never set by the product callers): 1 = no transposed LDS atomics,
 * 2 = no LDS traffic at all, 3 = LDS windows only (no matrix stream),
 * 4 = matrix stream only (no x gather / y flush)                             */
#define CFS_HIP_FLAG_ABLATE_MASK 7
/* keep rows in their given order (tiles = runs of consecutive rows) instead of
 * clustering the matrix graph first (the default, fewer halo columns)        */
#define CFS_HIP_FLAG_NO_REORDER 8
/* always keep the clustered order (skip the halo-count comparison)            */
#define CFS_HIP_FLAG_FORCE_CLUSTER 16
/* tune() of a matrix with >= 2M stored nonzeros measures alternatives and keeps the
 * fastest: when no block / slot count is given and the rows are scheduled in
 * clustered order, the default window shape (512 threads x 2 workgroups per CU)
 * against 1024 threads x 1 with a window twice the size; when a noticeable share of
 * the halo columns is used only once, the HYB form against the plain one.  This
 * flag skips the measurements (one schedule build; Tuning::None).              */
#define CFS_HIP_FLAG_NO_CALIBRATE 32
/* Shards only.  Default: off-block entries are MIRRORED -- stored by both ranks
 * they touch and processed one-sided, so that no contribution to y ever leaves
 * the rank and a sharded SpMV needs no exchange (x is replicated anyway; the
 * duplicated boundary entries are a few per cent of a shard).  With this flag a
 * shard keeps its off-block entries two-sided and packs its contributions to
 * rows of lower ranks for an all-to-all / reduce-scatter (the exchange form). */
#define CFS_HIP_FLAG_SHARD_EXCHANGE 64
/* Format::hyb (the reference's split_by_bandwidth, csr_matrix.tpp:312-401, as a
 * working feature): an entry whose column lies outside its tile and is used by that
 * tile only once leaves the symmetric tile format.  It is stored by BOTH tiles it
 * touches as (value, own row slot, global column) and processed one-sided with x
 * gathered from global memory -- no LDS slot, no strip entry, no fold entry.  With
 * the flag the split is always made; without it tune() may still choose it by
 * measurement (see CFS_HIP_FLAG_NO_CALIBRATE); CFS_HIP_FLAG_NO_HYB forbids it.     */
#define CFS_HIP_FLAG_HYB 128
#define CFS_HIP_FLAG_NO_HYB 256
/* Bit-reproducible results.  By default the transposed updates y_j += a_ij x_i of a
 * tile meet in LDS through floating-point atomics, whose order varies from run to
 * run: results agree to ~1e-13 of the row scale, not bitwise (the reference adds in a
 * fixed order, csr_matrix.tpp:3005-3013).  With this flag every contribution is
 * converted to a fixed-point number (2 x 40 bits below a PER-SLOT scale: the 1-norm of
 * the slot's matrix row times the largest |x| of the tile's window, a bound of every
 * partial sum) and accumulated with INTEGER LDS atomics -- associative, hence the same
 * bits whatever the order; the halo fold already adds in a fixed order.  The precision
 * does not depend on how the matrix is scaled.  Costs LDS (26 instead of 16 bytes per
 * slot: smaller tiles) and ALU; 512- or 1 024-thread workgroups.  Far entries (HYB) are
 * covered: the x values they gather from outside the window enter the tile's scale.  A
 * contribution keeps 2^-80 of (row 1-norm) x (largest |x| the TILE reads).  A NaN / Inf in x or in the matrix reads
 * NaN in the rows it reaches.  Same tolerance against the oracle as the default.   */
#define CFS_HIP_FLAG_DETERMINISTIC 1024
/* keep, for every stored value of the device format, its position in the caller's CSR
 * value array (4 bytes per stored nonzero of device memory): cfs_hip_sym_update_values_*
 * can then refresh the numbers of the matrix without repeating tune()             */
#define CFS_HIP_FLAG_KEEP_VALUE_MAP 2048
/* tune() builds the tile schedule ON THE GPU from one upload of the caller's CSR (split,
 * tile cut, slot tables, virtual rows, leaders, packing: HIP kernels) whenever
 * the input is covered by the device builder -- everything but rows that are unsorted, hold
 * duplicates or exceed 4096 stored entries (noticed on the GPU, cfs_hip_sym_debug_plan_note
 * says which).  With this flag the host builder
 * (cfs_plan.hpp, OpenMP) is used instead; the two produce the same schedule bit for bit
 * (cfs_hip_sym_debug_digest).                                                          */
#define CFS_HIP_FLAG_HOST_PLAN 4096

/* ---- tune() for a symmetric matrix
 *      (replaces CSRMatrix::tune -> compress_symmetry ->
 *       conflict_free_aposteriori, csr_matrix.tpp:230-310, :1204-1639).
 * Input is the FULL CSR exactly as CSRMatrix holds it before tune()
 * (csr_matrix.tpp:74-107: 0-based, rows ascending, columns ascending), in host
 * memory; the handle copies what it needs (the caller may free the arrays
 * afterwards, as compress_symmetry does at :1700-1706).  Only entries with
 * col <= row are read; a missing diagonal entry counts as 0.
 *
 * The *_shard_* forms build rank `rank`'s 1-D row block of a matrix sharded
 * over `nranks` GPUs at row boundaries row_splits[0..nranks].  By
 * default the shard is MIRRORED (see CFS_HIP_FLAG_SHARD_EXCHANGE): it also reads
 * the entries of its rows right of the block and the lower entries (c, i) they
 * mirror, its SpMV needs no exchange, and a matrix whose off-block structure is
 * not symmetric (or has duplicate entries there) is refused with an error whose
 * message starts with "mirror:" -- build it with CFS_HIP_FLAG_SHARD_EXCHANGE.   */
int cfs_hip_sym_create_f64(int n, const int *rowptr, const int *colind,
                           const double *values, const cfs_hip_options *opt,
                           cfs_hip_sym_t *out);
int cfs_hip_sym_create_f32(int n, const int *rowptr, const int *colind,
                           const float *values, const cfs_hip_options *opt,
                           cfs_hip_sym_t *out);
int cfs_hip_sym_create_shard_f64(int n, const int *rowptr, const int *colind,
                                 const double *values, int nranks, int rank,
                                 const int *row_splits,
                                 const cfs_hip_options *opt, cfs_hip_sym_t *out);
int cfs_hip_sym_create_shard_f32(int n, const int *rowptr, const int *colind,
                                 const float *values, int nranks, int rank,
                                 const int *row_splits,
                                 const cfs_hip_options *opt, cfs_hip_sym_t *out);
/* One host thread, `ngpus` GPUs (the C++ surface with CFS_NUM_GPUS; the reference's
 * knob of this kind is CFS_NUM_THREADS, src/runtime.cpp:10-21): ngpus mirrored row
 * blocks, shard g on devices[g] (NULL: the visible devices round-robin from the
 * current one; several shards may share a device), each on a stream of its own.
 * The handle behaves like a whole-matrix one: cfs_hip_sym_spmv[_async] take x / y
 * of n entries on the device that was current at create (or host pointers);
 * shards on other devices get a replica of x per SpMV and copy their y block back
 * (cfs_hip_sym_multi_set_xmode; or reach both through peer access).            */
int cfs_hip_sym_create_multi_f64(int n, const int *rowptr, const int *colind,
                                 const double *values, int ngpus, const int *devices,
                                 const cfs_hip_options *opt, cfs_hip_sym_t *out);
int cfs_hip_sym_create_multi_f32(int n, const int *rowptr, const int *colind,
                                 const float *values, int ngpus, const int *devices,
                                 const cfs_hip_options *opt, cfs_hip_sym_t *out);
int cfs_hip_sym_num_gpus(cfs_hip_sym_t h, int *ngpus); /* shards of the handle (1: plain) */
/* How the shards of a multi-device handle that live on ANOTHER device than its home reach x / y.
 * REPLICATE (default; env CFS_MULTI_X=replicate): x is replicated -- one peer copy home ->
 * device per shard and SpMV, kernels gather x and write their y block in local HBM, one peer
 * copy brings the block home (north_star: "x replicated").  PEER (CFS_MULTI_X=peer): kernels
 * read x / write y in the home device's memory through peer access over xGMI, no copies.
 * REPLICATE_ALL copies for shards on the home device too (tests on a one-GPU box).
 * NOTE: with more than one physical device neither form has run on hardware yet (the build
 * and test boxes have one GPU); treat cross-device operation as unverified.           */
#define CFS_HIP_XMODE_PEER 0
#define CFS_HIP_XMODE_REPLICATE 1
#define CFS_HIP_XMODE_REPLICATE_ALL 2
int cfs_hip_sym_multi_set_xmode(cfs_hip_sym_t h, int xmode);
/* devices[g] = device of shard g (may be NULL); *distinct = number of distinct devices */
int cfs_hip_sym_multi_devices(cfs_hip_sym_t h, int *devices, int capacity, int *distinct);
/* nnz_low-balanced row boundaries (multiples of 16, csr_matrix.tpp:418) for
 * sharding; row_splits has nranks+1 entries.                                 */
int cfs_hip_sym_balanced_splits(int n, const int *rowptr, const int *colind,
                                int nranks, int *row_splits);
int cfs_hip_sym_destroy(cfs_hip_sym_t h);
/* New values, SAME sparsity pattern (a stiffness matrix reassembled in every Newton step;
 * the reference would run tune() -- split, conflict graph, colouring,
 * csr_matrix.tpp:1204-1639 -- again): `values` is the full CSR value array in the order
 * of the rowptr / colind the handle was created from, nnz entries, host or device
 * pointer.  A device kernel pours it into the existing schedule (the value half of
 * tune()'s packing, on the GPU); tiles, slots, fold index stay.  The handle must have
 * been created with CFS_HIP_FLAG_KEEP_VALUE_MAP; returns after the new values are in
 * place.  Not for CFS_HIP_FLAG_DETERMINISTIC handles (their scale depends on the values). */
int cfs_hip_sym_update_values_f64(cfs_hip_sym_t h, const double *values, long long nnz);
int cfs_hip_sym_update_values_f32(cfs_hip_sym_t h, const float *values, long long nnz);

/* ---- dense_vector_multiply (replaces spmv_fn = cpu_mv_sym_conflict_free_v2,
 *      csr_matrix.tpp:2965-3028).  y is fully overwritten (the reference test
 *      never zeroes it: test/test_spmv_mmf.cpp:71,82-83); x and y must not
 *      alias.  x has n entries, y has n entries (block rows for a shard).
 *
 * cfs_hip_sym_spmv      : x / y may be host or device pointers (detected).
 *                         With a host pointer the call stages through PCIe and
 *                         returns after the result is complete.  With both
 *                         vectors device-resident it is enqueued on the
 *                         library stream and returns at once; cfs_hip_memcpy
 *                         and cfs_hip_synchronize(NULL) wait for that stream,
 *                         so a caller that reads y back always sees it done.
 * cfs_hip_sym_spmv_async: device pointers only, enqueued on `stream`
 *                         (a hipStream_t; NULL = HIP's null stream), returns
 *                         immediately.                                       */
int cfs_hip_sym_spmv(cfs_hip_sym_t h, void *y, const void *x);
/* A solver-style caller of the path, native (no counterpart in the reference: its only callers
 * are a benchmark loop and a self-check with a fixed x, bench/bench_spmv_mmf.cpp:139-173):
 * conjugate gradients for A u = b on RESIDENT vectors of a symmetric positive definite matrix.
 * u_dev: in = first guess, out = solution; b_dev: right-hand side; both device pointers of the
 * handle's device and value type, 16-byte aligned.  An iteration is five launches on `stream` -- the SpMV (two),
 * p.q, the fused update of u and r with r.r, the new direction -- with every scalar in device
 * memory: no host round trip inside the loop; the host reads the convergence flag every
 * `check_every` iterations (<= 0: 8; at most 16).  Stops when ||r|| <= tol ||b|| (the recurrence's r) or after
 * maxiter iterations; *iterations = iterations done, *relres = ||b - A u|| / ||b|| RECOMPUTED from
 * the returned u.  Dot products are accumulated in fp64.  Returns after the result is complete.
 * A handle of the whole matrix: one device, or a multi-device handle (cfs_hip_sym_create_multi_*:
 * the vector kernels run on its home device, the products on all of them); not a shard.       */
int cfs_hip_sym_cg(cfs_hip_sym_t h, void *u_dev, const void *b_dev, double tol, int maxiter, int check_every,
                   int *iterations, double *relres, void *stream);
int cfs_hip_sym_spmv_async(cfs_hip_sym_t h, void *y_dev, const void *x_dev,
                           void *stream);

/* ---- sharded operation: y_block = local rows; contributions to rows owned
 *      by lower ranks are packed into send_buf (device), exchanged by the
 *      caller (RCCL all-to-all / reduce-scatter), and folded in by
 *      cfs_hip_sym_recv_fold.                                                */
/* send_counts[r] = number of packed values destined to rank r (0 for r>=rank) */
int cfs_hip_sym_shard_send_counts(cfs_hip_sym_t h, int *send_counts);
/* global row index of every packed value, in send order (host array)        */
int cfs_hip_sym_shard_send_rows(cfs_hip_sym_t h, int *rows);
/* describe what this rank will receive: recv_rows are global row indices in
 * receive-buffer order (concatenated by source rank), host array            */
int cfs_hip_sym_shard_set_recv(cfs_hip_sym_t h, int nrecv, const int *recv_rows);
int cfs_hip_sym_spmv_local_async(cfs_hip_sym_t h, void *y_block_dev,
                                 const void *x_dev, void *send_buf_dev,
                                 void *stream);
int cfs_hip_sym_recv_fold_async(cfs_hip_sym_t h, void *y_block_dev,
                                const void *recv_buf_dev, void *stream);
/* The SpMV of a handle is two launches -- the tile kernel (streams the matrix;
 * the roofline kernel) and the halo fold -- plus, for a shard, the pack of the
 * contributions to lower ranks.  This entry point enqueues only the selected
 * ones (in the order tiles, pack, fold) so that bench.py can bracket the tile
 * kernel with HIP events and a shard can start its exchange before the local
 * fold; CFS_HIP_PHASE_ALL is a full local SpMV.                               */
#define CFS_HIP_PHASE_TILES 1
#define CFS_HIP_PHASE_FOLD 2
#define CFS_HIP_PHASE_PACK 4
#define CFS_HIP_PHASE_ALL 7
int cfs_hip_sym_spmv_phases_async(cfs_hip_sym_t h, void *y_block_dev,
                                  const void *x_dev, void *send_buf_dev,
                                  int phases, void *stream);

/* ---- native exchange between the row blocks of one process's devices: the north-star's
 *      reduce-scatter over xGMI, and the y -> x all-gather of a solver loop, without Python.
 *      (The reference has no exchange beyond its barrier between colours,
 *      csr_matrix.tpp:3018; what crosses a block boundary here are its direct conflicts,
 *      :1443-1451.)  Transports: RCCL (one communicator per device, ncclCommInitAll;
 *      librccl.so is loaded at the first use, the library does not link against it) and PEER
 *      (plain kernels / copies over peer access: what several ranks on ONE device -- the
 *      test boxes -- use, RCCL refusing two ranks per device; also the fall-back when RCCL
 *      cannot be loaded).  AUTO = RCCL when the devices are distinct and it loads.
 *      NOTE: with more than one rank the RCCL transport has not run on hardware yet.      */
typedef struct cfs_hip_comm_s *cfs_hip_comm_t;
#define CFS_HIP_TRANSPORT_AUTO 0
#define CFS_HIP_TRANSPORT_RCCL 1
#define CFS_HIP_TRANSPORT_PEER 2
/* ranks 0..ndev-1 on devices[] (NULL: the visible devices round-robin from the current one) */
int cfs_hip_comm_create(int ndev, const int *devices, int transport, cfs_hip_comm_t *out);
int cfs_hip_comm_info(cfs_hip_comm_t c, int *ndev, int *transport);
int cfs_hip_comm_destroy(cfs_hip_comm_t c);
/* sum-reduce-scatter: rank g contributes send[g] (ndev * count values, on its device) and
 * receives the g-th block of the sum in recv[g] (count values); enqueued on streams[g]     */
int cfs_hip_comm_reduce_scatter(cfs_hip_comm_t c, void *const *send, void *const *recv, size_t count,
                                int value_bytes, void *const *streams);
/* all-gather: recv[g] (ndev * count values) = the blocks send[0..ndev-1] (count values each) */
int cfs_hip_comm_allgather(cfs_hip_comm_t c, void *const *send, void *const *recv, size_t count,
                           int value_bytes, void *const *streams);
/* before rank `rank` overwrites its send buffer on `stream`: wait until the previous
 * collective has consumed it (PEER transport: other ranks' kernels read it)                */
int cfs_hip_comm_wait_consumed(cfs_hip_comm_t c, int rank, void *stream);

/* ---- introspection (A->nnz(), A->size(), and what bench.py needs) --------- */
typedef struct {
  int n;               /* matrix order                                    */
  int row_begin, row_end; /* rows owned by this handle                    */
  int value_bytes;     /* 8 or 4                                          */
  int64_t nnz_low;     /* stored strict-lower nonzeros of the owned rows  */
  int64_t nnz_diag;    /* stored diagonal entries                         */
  int64_t nnz_full;    /* expanded count = what A->nnz() reports          */
  int ntiles, nslices, max_slots_used, block_threads;
  int64_t halo_slots;  /* sum over tiles of halo (non-own) slots          */
  int64_t fold_rows;   /* destination rows touched by the halo fold       */
  int64_t remote_vals; /* packed values sent to lower ranks (shards)      */
  int64_t lds_bytes;   /* dynamic LDS per workgroup                       */
  /* algorithmic bytes of one SpMV over the owned rows, SURVEY.md 8(d):
   * nnz_low*(4+s) + rows*(4+3s)                                          */
  int64_t bytes_algorithmic;
  /* bytes the device format actually streams per SpMV (values + 16-bit
   * slots + per-row metadata + x/y + halo strips + fold index)           */
  int64_t bytes_streamed;
  int64_t device_bytes; /* device memory held by the handle               */
  int64_t mirror_entries; /* one-sided entries a mirrored shard stores for rows of
                             higher ranks (0 for a whole matrix / exchange form)  */
  int64_t far_entries; /* HYB: nonzeros kept outside the tile format (each stored twice) */
  int ngroups;         /* persistent workgroups of a launch                          */
  int reserved_;
} cfs_hip_sym_stats;
int cfs_hip_sym_get_stats(cfs_hip_sym_t h, cfs_hip_sym_stats *out);
/* developer diagnostic: one extra launch of the tile kernel that records, per
 * persistent workgroup (in blockIdx order), 8 words of 100 MHz wall-clock
 * stamps: [0] start, [1] first x window ready (after the barrier), [2] slices
 * of the last tile done, [3] end, [4] first tile descriptor loaded, [5] slot
 * table of the first tile arrived (thread 0), [6] its x values arrived
 * (thread 0), [7] wave 0 finished its last slice.  Not used by any product
 * path; tools/timeline.py reads it.                                          */
int cfs_hip_sym_debug_timeline(cfs_hip_sym_t h, void *y_dev, const void *x_dev,
                               unsigned long long *stamps, int capacity_words,
                               int *ngroups);

/* developer diagnostic: what the group in every launch slot (block b runs slot
 * (b % 8) * (ngroups / 8) + b / 8) has to do, CFS_HIP_GROUP_FEATURES words
 * each: [0] tiles, [1] rows, [2] virtual rows, [3] slices, [4] packet rounds (sum
 * over slices of the longest lane's packets), [5] value-stream entries, [6] slot-
 * stream entries, [7] COO leftovers, [8] halo slots, [9] slots.  tools/ fit the
 * cost model of the row cut against the timeline with it.                      */
#define CFS_HIP_GROUP_FEATURES 10
int cfs_hip_sym_debug_group_features(cfs_hip_sym_t h, long long *out, int capacity_words,
                                     int *ngroups);

/* developer / test: 64-bit FNV-1a digests of the device arrays of a handle's schedule, over
 * their logical lengths, in this order: tiles (aexp masked), launch-slot first tiles, launch-
 * slot tile ranges, slot_col, rowinfo, diag, slice_meta, leadlane, vals, slots, cvals, crows,
 * ccols, fold records, fold remainder lists, val_map, cval_map, diag_map, window / launch shape,
 * slot exponents (deterministic build), send_ptr, send_idx (exchange-form shard), the far sections
 * (fvals, frows, fcols, fval_map, count) -- 0 when absent.  A
 * schedule built on the GPU and one built by the host builder for the same matrix and options
 * have equal digests.  words[CFS_HIP_DIGEST_WORDS - 1] = 1 if the handle's schedule was built
 * on the GPU.                                                                           */
#define CFS_HIP_DIGEST_WORDS 28
int cfs_hip_sym_debug_digest(cfs_hip_sym_t h, unsigned long long *words, int capacity_words);
/* why the device builder handed this handle's schedule to the host builder ("" = it built it) */
int cfs_hip_sym_debug_plan_note(cfs_hip_sym_t h, char *buf, int capacity);

/* ---- host-only self-check of the tile schedule (needs no GPU): builds the
 *      schedule tune() would upload, decodes it back to (row, col, value)
 *      triples and compares them with the strict lower triangle of the input;
 *      also checks that the halo-fold index covers every strip entry once.
 *      Structure only -- no SpMV arithmetic is performed on the host.      ---- */
typedef struct {
  int ntiles, ngroups, lds_slots;
  int64_t nslices, halo_slots, stream_len, nnz_low, fold_rows, remote_vals;
  int64_t decoded;    /* triples recovered from the device format           */
  int64_t mismatches; /* 0 = the schedule encodes exactly the input         */
  int64_t mirror_entries; /* mirrored off-block entries (shards, default form)  */
  int64_t far_entries;    /* HYB: nonzeros outside the tile format (mirror images checked) */
} cfs_hip_plan_report;
int cfs_hip_sym_plan_check_f64(int n, const int *rowptr, const int *colind,
                               const double *values, int nranks, int rank,
                               const int *row_splits, const cfs_hip_options *opt,
                               cfs_hip_plan_report *report);
int cfs_hip_sym_plan_check_f32(int n, const int *rowptr, const int *colind,
                               const float *values, int nranks, int rank,
                               const int *row_splits, const cfs_hip_options *opt,
                               cfs_hip_plan_report *report);

/* host-only: the send side of rank `rank`'s shard (what cfs_hip_sym_shard_send_
 * counts / _rows would return) without touching a device; used by the CPU
 * (gloo) tests of the exchange.  rows may be NULL to query *nrows_out.        */
int cfs_hip_sym_plan_send_info_f64(int n, const int *rowptr, const int *colind,
                                   const double *values, int nranks, int rank,
                                   const int *row_splits, const cfs_hip_options *opt,
                                   int *send_counts, int *rows, int rows_cap,
                                   int *nrows_out);

/* ---- general CSR (replaces cpu_mv / cpu_mv_serial, csr_matrix.tpp:2664-2704:
 *      Format::csr and the silent fall-back for non-symmetric files,
 *      csr_matrix.tpp:13-19)                                              ---- */
int cfs_hip_csr_create_f64(int nrows, int ncols, const int *rowptr,
                           const int *colind, const double *values,
                           cfs_hip_csr_t *out);
int cfs_hip_csr_create_f32(int nrows, int ncols, const int *rowptr,
                           const int *colind, const float *values,
                           cfs_hip_csr_t *out);
int cfs_hip_csr_spmv(cfs_hip_csr_t h, void *y, const void *x);
int cfs_hip_csr_spmv_async(cfs_hip_csr_t h, void *y_dev, const void *x_dev,
                           void *stream);
int cfs_hip_csr_destroy(cfs_hip_csr_t h);
/* The general CSR kernel exists in two forms -- a workgroup per block of rows (products staged
 * in LDS between two barriers) and a wave per chunk of rows (no workgroup barrier, the next
 * chunk's loads in flight) -- within a few per cent of each other, the order depending on the
 * matrix and the box: the FIRST SpMV of a handle with >= 1M nonzeros times five SpMVs of each
 * (y is fully overwritten by either) and keeps the faster.  CFS_HIP_CSR_KERNEL=block|wave
 * pins the form.                                                                        */
#define CFS_HIP_CSR_FORM_BLOCK 0
#define CFS_HIP_CSR_FORM_WAVE 1
int cfs_hip_csr_kernel_form(cfs_hip_csr_t h, int *form, int *measured);
/* The block form reads 16-bit column codes (2 bytes per nonzero instead of 4) in every row
 * block whose columns fit into four windows of 16 384 columns (window << 14 | offset; banded
 * matrices, 3-D stencils; written once, on the device, when the handle is created;
 * CFS_HIP_CSR_COL16=0: never); the codes -- or, for a block that needs more windows, its 32-bit
 * columns -- and a copy of the block's values are stored in the order the lanes consume them; and
 * both forms hand the k-th eighth of the matrix to XCD k (CFS_HIP_CSR_XCD=0: round robin).
 * bytes_streamed = bytes one SpMV of the kept form reads from the handle's arrays;
 * narrow_nnz = nonzeros stored with 16-bit columns.                                          */
int cfs_hip_csr_stats(cfs_hip_csr_t h, int64_t *bytes_streamed, int64_t *narrow_nnz);

/* ---- HIP-event timing on the stream the kernels run on (bench.py) --------- */
int cfs_hip_event_create(void **ev);
int cfs_hip_event_record(void *ev, void *stream);
int cfs_hip_event_elapsed_ms(void *start, void *stop, float *ms); /* syncs stop */
int cfs_hip_event_destroy(void *ev);

#ifdef __cplusplus
}
#endif
#endif /* CFS_HIP_H */
