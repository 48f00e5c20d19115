/*
 * cfs_oracle.c -- CPU oracle (plain C restatement of athelaf/cfs-spmv).
 *
 * TEST INFRASTRUCTURE ONLY -- see cfs_oracle.h for who may load it and for the
 * pinning status ("parity unpinned" for the floating-point kernels; the MMF
 * reader / CSR structure is pinned against the genuine reference reader built
 * into oracle/_ref/).
 *
 * Reference paths are relative to the reference root (include/..., src/...).
 */
#define _GNU_SOURCE
#include "cfs_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[256];
const char *orc_last_error(void) { return g_err; }
void orc_free(void *p) { free(p); }

static int fail(int code, const char *msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}

/* 64-byte aligned allocation, src/allocator.cpp:24-43 */
static void *xalloc(size_t bytes) {
  void *p = NULL;
  if (bytes == 0) bytes = 64;
  if (posix_memalign(&p, 64, bytes) != 0) {
    fprintf(stderr, "[oracle] posix_memalign(%zu) failed\n", bytes);
    abort();
  }
  return p;
}

/* src/runtime.cpp:10-21: unset -> 1, negative -> 1, "0" stays 0 */
int orc_get_num_threads(void) {
  const char *e = getenv("CFS_NUM_THREADS");
  int ret = 1;
  if (e) {
    ret = atoi(e);
    if (ret < 0) ret = 1;
  }
  return ret;
}

/* include/utils/platform.hpp:27-37 */
int orc_is_equal_f64(double x, double y) {
  return fabs(x - y) <= 1e-8 * fabs(x);
}
int orc_is_equal_f32(float x, float y) {
  const float epsilon = 1e-4f;
  return fabsf(x - y) <= epsilon * fabsf(x);
}

/* =========================================================================
 * a1: Matrix-Market reader (include/io/mmf.hpp, src/mmf.cpp)
 * ========================================================================= */

typedef struct {
  char **tok;
  int ntok;
  char *buf; /* owns the token storage */
} args_t;

static void args_clear(args_t *a) {
  free(a->tok);
  free(a->buf);
  a->tok = NULL;
  a->buf = NULL;
  a->ntok = 0;
}

/* DoRead, src/mmf.cpp:26-44: getline; a line that ends at EOF without '\n'
 * sets eofbit and is reported as "no line" (so a last line needs its '\n');
 * leading/trailing blanks and tabs are trimmed; tokens are split on a single
 * ' ' only (split(), src/mmf.cpp:6-23: empty tokens are dropped, tabs inside
 * the line do NOT separate).  Returns 1 if a line was read, 0 at EOF.        */
static int do_read(FILE *in, args_t *a) {
  char *line = NULL;
  size_t cap = 0;
  ssize_t len = getline(&line, &cap, in);
  if (len < 0 || line[len - 1] != '\n') { /* eofbit set -> false */
    free(line);
    return 0;
  }
  line[--len] = '\0';
  /* trim " \t" on both ends */
  char *s = line;
  while (*s == ' ' || *s == '\t') s++;
  char *e = line + len;
  while (e > s && (e[-1] == ' ' || e[-1] == '\t')) e--;
  *e = '\0';
  args_clear(a);
  a->buf = strdup(s);
  free(line);
  int maxtok = 1;
  for (char *p = a->buf; *p; p++)
    if (*p == ' ') maxtok++;
  a->tok = (char **)malloc(sizeof(char *) * (size_t)maxtok);
  char *p = a->buf;
  while (*p) {
    while (*p == ' ') *p++ = '\0';
    if (!*p) break;
    a->tok[a->ntok++] = p;
    while (*p && *p != ' ') p++;
  }
  return 1;
}

typedef struct {
  int row, col; /* one-based, as the reference's Elem keeps them */
  double val;
  long seq; /* input order: tie-break => a deterministic stable sort */
} elem_t;

/* ElemSorter, include/io/mmf.hpp:22-38 ((row, col) ascending).  The reference
 * uses std::sort, whose order among duplicates is unspecified; the oracle
 * fixes it to input order.                                                   */
static int elem_cmp(const void *pa, const void *pb) {
  const elem_t *a = (const elem_t *)pa, *b = (const elem_t *)pb;
  if (a->row != b->row) return a->row < b->row ? -1 : 1;
  if (a->col != b->col) return a->col < b->col ? -1 : 1;
  return a->seq < b->seq ? -1 : (a->seq > b->seq);
}

/* ParseElement, include/io/mmf.hpp:327-343 */
static int parse_element(const args_t *a, int *y, int *x, double *v) {
  if (a->ntok >= 3) {
    *y = atoi(a->tok[0]);
    *x = atoi(a->tok[1]);
    *v = atof(a->tok[2]);
  } else if (a->ntok == 2) {
    *y = atoi(a->tok[0]);
    *x = atoi(a->tok[1]);
    *v = 0.42; /* pattern entries, mmf.hpp:334-337 */
  } else {
    return -1; /* "bad input, less arguments in line of MMF file." */
  }
  return 0;
}

typedef struct {
  int nrows, ncols, nnz_decl;
  int symmetric, col_wise, zero_based;
  elem_t *elems;
  long nelems;
} mmf_t;

static int mmf_read(const char *path, mmf_t *m) {
  memset(m, 0, sizeof *m);
  m->col_wise = 1; /* mmf.hpp:181 */
  FILE *in = fopen(path, "r");
  if (!in) return fail(-1, "MMF file error.");
  args_t a = {0};
  int file_mode = 0, rc = 0;

  if (!do_read(in, &a) || a.ntok == 0) {
    rc = fail(-2, "empty MMF file");
    goto done;
  }
  /* ParseMmfHeaderLine, mmf.hpp:203-251 */
  if (strcmp(a.tok[0], "%%MatrixMarket") != 0) {
    if (strlen(a.tok[0]) > 2 && a.tok[0][0] == '%' && a.tok[0][1] == '%') {
      rc = fail(-3, "invalid header line in MMF file.");
      goto done;
    }
    file_mode = 1; /* no banner: first line is (maybe) the size line */
  } else {
    if (a.ntok < 5) {
      rc = fail(-4, "less arguments in header line of MMF file.");
      goto done;
    }
    if (strcmp(a.tok[2], "coordinate") != 0) {
      rc = fail(-5, "unsupported matrix format in header line of MMF file.");
      goto done;
    }
    /* the field token (real/integer/pattern) is ignored, mmf.hpp:225-237 */
    if (strcmp(a.tok[4], "general") == 0)
      m->symmetric = 0;
    else if (strcmp(a.tok[4], "symmetric") == 0)
      m->symmetric = 1;
    else {
      rc = fail(-6, "unsupported symmetry in header line of MMF file.");
      goto done;
    }
    for (int i = 5; i < a.ntok; i++) {
      if (!strcmp(a.tok[i], "base-0")) m->zero_based = 1;
      else if (!strcmp(a.tok[i], "base-1")) m->zero_based = 0;
      else if (!strcmp(a.tok[i], "column")) m->col_wise = 1;
      else if (!strcmp(a.tok[i], "row")) m->col_wise = 0;
    }
  }
  /* ParseMmfSizeLine, mmf.hpp:254-272: '%' lines are skipped only here */
  {
    int ignore_comments = file_mode && a.tok[0][0] == '%';
    if (!file_mode || ignore_comments) {
      int c;
      while ((c = fgetc(in)) == '%') {
        while ((c = fgetc(in)) != EOF && c != '\n') {
        }
      }
      if (c != EOF) ungetc(c, in);
      if (!do_read(in, &a)) {
        rc = fail(-7, "size line error in MMF file.");
        goto done;
      }
    }
    double v;
    if (a.ntok == 0 || parse_element(&a, &m->nrows, &m->ncols, &v) != 0) {
      rc = fail(-8, "bad input, less arguments in line of MMF file.");
      goto done;
    }
    /* nr_nzeros_ is an IndexType: ParseElement deduces ValueType=int */
    m->nnz_decl = (int)v;
  }
  /* DoLoadMmfMatrix, mmf.hpp:275-306 (symmetric_ || col_wise_), or the
   * streaming iterator for "row" general files (mmf.hpp:117-169)           */
  {
    long cap = (long)m->nnz_decl * (m->symmetric ? 2 : 1);
    if (cap < 1) cap = 1;
    m->elems = (elem_t *)malloc(sizeof(elem_t) * (size_t)cap);
    for (int i = 0; i < m->nnz_decl; i++) {
      elem_t e;
      if (!do_read(in, &a)) {
        rc = fail(-9, "Requesting dereference, but mmf ended.");
        goto done;
      }
      if (a.ntok == 0 || parse_element(&a, &e.row, &e.col, &e.val) != 0) {
        rc = fail(-8, "bad input, less arguments in line of MMF file.");
        goto done;
      }
      if (m->zero_based) {
        e.row++;
        e.col++;
      }
      e.seq = m->nelems;
      m->elems[m->nelems++] = e;
      if (m->symmetric && e.row != e.col) {
        int t = e.row;
        e.row = e.col;
        e.col = t;
        e.seq = m->nelems;
        m->elems[m->nelems++] = e;
      }
    }
    if (m->symmetric || m->col_wise)
      qsort(m->elems, (size_t)m->nelems, sizeof(elem_t), elem_cmp);
  }
done:
  args_clear(&a);
  fclose(in);
  if (rc != 0) {
    free(m->elems);
    m->elems = NULL;
  }
  return rc;
}

/* a2: CSRMatrix(filename, ...) ctor, include/matrix/csr_matrix.tpp:8-111.
 * COO (1-based, sorted) -> CSR (0-based); empty rows repeat rowptr (:91-96).
 * The asserts of :84-87,:104-105 become error returns.                      */
static int mmf_to_csr(const mmf_t *m, int **rowptr_out, int **colind_out,
                      double **val_out) {
  int n = m->nrows;
  long nnz = m->nelems;
  int *rowptr = (int *)xalloc(sizeof(int) * ((size_t)n + 1));
  int *colind = (int *)xalloc(sizeof(int) * (size_t)nnz);
  double *values = (double *)xalloc(sizeof(double) * (size_t)nnz);
  memset(rowptr, 0, sizeof(int) * ((size_t)n + 1));
  int row_i = 0, val_i = 0, row_prev = 0, rc = 0;
  rowptr[row_i++] = val_i;
  for (long k = 0; k < nnz; k++) {
    int row = m->elems[k].row - 1, col = m->elems[k].col - 1;
    if (row < row_prev || row >= n || col < 0 || col >= m->ncols) {
      rc = fail(-10, "CSR ctor assert: entry out of order or out of range");
      break;
    }
    if (row != row_prev) {
      for (int i = 0; i < row - row_prev; i++) rowptr[row_i++] = val_i;
      row_prev = row;
    }
    colind[val_i] = col;
    values[val_i] = m->elems[k].val;
    val_i++;
  }
  if (rc == 0) {
    rowptr[row_i] = val_i;
    if (row_i != n) /* assert(row_i == nrows_): last row must be non-empty */
      rc = fail(-11, "CSR ctor assert: row_i != nrows (empty trailing rows)");
  }
  if (rc != 0) {
    free(rowptr);
    free(colind);
    free(values);
    return rc;
  }
  *rowptr_out = rowptr;
  *colind_out = colind;
  *val_out = values;
  return 0;
}

int orc_mmf_load_f64(const char *path, int *nrows, int *ncols, int *nnz,
                     int *symmetric, int **rowptr, int **colind,
                     double **values) {
  mmf_t m;
  int rc = mmf_read(path, &m);
  if (rc) return rc;
  rc = mmf_to_csr(&m, rowptr, colind, values);
  if (rc == 0) {
    *nrows = m.nrows;
    *ncols = m.ncols;
    *nnz = (int)m.nelems;
    *symmetric = m.symmetric;
  }
  free(m.elems);
  return rc;
}

int orc_mmf_load_f32(const char *path, int *nrows, int *ncols, int *nnz,
                     int *symmetric, int **rowptr, int **colind,
                     float **values) {
  double *v64 = NULL;
  int rc = orc_mmf_load_f64(path, nrows, ncols, nnz, symmetric, rowptr, colind,
                            &v64);
  if (rc) return rc;
  /* `v = atof(..)` into a float ValueType: one double->float rounding */
  float *v32 = (float *)xalloc(sizeof(float) * (size_t)(*nnz));
  for (int i = 0; i < *nnz; i++) v32[i] = (float)v64[i];
  free(v64);
  *values = v32;
  return 0;
}

/* =========================================================================
 * a6: partitioning
 * ========================================================================= */

/* partition_by_nrows, csr_matrix.tpp:403-435.  BlkFactor = 16
 * (csr_matrix.hpp:89-90).  The reference does not clamp: for tiny n and large
 * T, (T-1)*per can exceed n and it then indexes out of range; the oracle
 * reports that case instead.                                                 */
int orc_partition_by_nrows(int nrows, int nthreads, int *row_split) {
  const int BlkFactor = 16;
  if (nthreads < 1) return fail(-20, "nthreads < 1");
  int per = ((nrows / nthreads - 1) | (BlkFactor - 1)) + 1;
  row_split[0] = 0;
  for (int i = 0; i < nthreads - 1; i++) row_split[i + 1] = row_split[i] + per;
  row_split[nthreads] = nrows;
  for (int i = 0; i < nthreads; i++)
    if (row_split[i] > row_split[i + 1])
      return fail(-21, "partition_by_nrows: (T-1)*per exceeds nrows "
                       "(undefined behaviour in the reference)");
  return 0;
}

/* partition_by_nnz, unsymmetric branch, csr_matrix.tpp:437-541 */
int orc_partition_by_nnz(int nrows, const int *rowptr, int nthreads,
                         int *row_split) {
  const int BlkFactor = 16;
  if (nthreads < 1) return fail(-20, "nthreads < 1");
  if (nthreads == 1) {
    row_split[0] = 0;
    row_split[1] = nrows;
    return 0;
  }
  int nnz_cnt = rowptr[nrows];
  int nnz_per_split = nnz_cnt / nthreads;
  int curr_nnz = 0, split_cnt = 0;
  row_split[0] = 0;
  for (int i = 0; i < nrows; i++) {
    curr_nnz += rowptr[i + 1] - rowptr[i];
    if (curr_nnz >= nnz_per_split && ((i + 1) % BlkFactor == 0)) {
      ++split_cnt;
      if (split_cnt <= nthreads) row_split[split_cnt] = i + 1;
      curr_nnz = 0;
    }
  }
  if (curr_nnz < nnz_per_split && split_cnt <= nthreads) {
    ++split_cnt;
    if (split_cnt <= nthreads) row_split[split_cnt] = nrows;
  }
  if (split_cnt > nthreads) row_split[nthreads] = nrows;
  for (int i = split_cnt + 1; i <= nthreads; i++) row_split[i] = nrows;
  return 0;
}

/* =========================================================================
 * conflict graph + colouring (value-type independent)
 * ========================================================================= */

typedef struct {
  int *adj_ptr; /* [V+1] */
  int *adj;     /* sorted, unique neighbour lists */
  int V;
} graph_t;

/* per-thread edge SET (the reference inserts into tbb concurrent sets, so an
 * edge found a million times is stored once; a plain list would need
 * O(rows * deg^2) memory at high thread counts): open addressing on the
 * packed undirected pair (min << 32 | max).                                  */
typedef struct {
  unsigned long *slot; /* 0 = empty (pair (0,0) is a self loop, never stored) */
  long n, cap;
  long *e; /* filled by eb_finish: both directions, packed (u << 32 | v) */
} edgebuf_t;

static void eb_insert_raw(edgebuf_t *b, unsigned long key) {
  unsigned long h = key * 0x9e3779b97f4a7c15UL;
  long i = (long)(h >> 20) & (b->cap - 1);
  while (b->slot[i] != 0) {
    if (b->slot[i] == key) return;
    i = (i + 1) & (b->cap - 1);
  }
  b->slot[i] = key;
  b->n++;
}

static void eb_push(edgebuf_t *b, int u, int v) {
  if (u == v) return; /* cannot happen, see graph_from_edges */
  if (b->cap == 0) {
    b->cap = 1024;
    b->slot = (unsigned long *)calloc((size_t)b->cap, sizeof(unsigned long));
  } else if (2 * (b->n + 1) > b->cap) {
    edgebuf_t nb = {0};
    nb.cap = b->cap * 2;
    nb.slot = (unsigned long *)calloc((size_t)nb.cap, sizeof(unsigned long));
    for (long i = 0; i < b->cap; i++)
      if (b->slot[i]) eb_insert_raw(&nb, b->slot[i]);
    free(b->slot);
    b->slot = nb.slot;
    b->cap = nb.cap;
    b->n = nb.n;
  }
  unsigned long lo = (unsigned long)(u < v ? u : v), hi = (unsigned long)(u < v ? v : u);
  eb_insert_raw(b, (lo << 32) | hi);
}

/* expand the set into the directed pair list graph_from_edges consumes */
static void eb_finish(edgebuf_t *b) {
  long m = 0;
  b->e = (long *)malloc(sizeof(long) * (size_t)(2 * b->n + 1));
  for (long i = 0; i < b->cap; i++)
    if (b->slot[i]) {
      unsigned long lo = b->slot[i] >> 32, hi = b->slot[i] & 0xffffffffUL;
      b->e[m++] = (long)((lo << 32) | hi);
      b->e[m++] = (long)((hi << 32) | lo);
    }
  free(b->slot);
  b->slot = NULL;
  b->n = m;
}

static int long_cmp(const void *a, const void *b) {
  long x = *(const long *)a, y = *(const long *)b;
  return x < y ? -1 : (x > y);
}

/* g[u].insert(v) / g[v].insert(u): build sorted-unique adjacency lists from
 * the per-thread edge buffers (the reference uses tbb concurrent sets,
 * csr_matrix.hpp:83-84; iteration order of a set does not influence the
 * colouring because neighbour colours are only *marked*, :2048-2049).
 * Self loops cannot occur: every row_split_ entry but the last is a multiple
 * of BlkFactor, so two rows on different threads never share a 16-row block. */
static void graph_from_edges(graph_t *g, int V, edgebuf_t *bufs, int nbufs) {
  long total = 0;
  for (int i = 0; i < nbufs; i++) total += bufs[i].n;
  long *all = (long *)malloc(sizeof(long) * (size_t)(total ? total : 1));
  long k = 0;
  for (int i = 0; i < nbufs; i++) {
    memcpy(all + k, bufs[i].e, sizeof(long) * (size_t)bufs[i].n);
    k += bufs[i].n;
  }
  qsort(all, (size_t)total, sizeof(long), long_cmp);
  long uniq = 0;
  for (long i = 0; i < total; i++)
    if (i == 0 || all[i] != all[i - 1]) all[uniq++] = all[i];
  g->V = V;
  g->adj_ptr = (int *)calloc((size_t)V + 1, sizeof(int));
  g->adj = (int *)malloc(sizeof(int) * (size_t)(uniq ? uniq : 1));
  for (long i = 0; i < uniq; i++) g->adj_ptr[(all[i] >> 32) + 1]++;
  for (int v = 0; v < V; v++) g->adj_ptr[v + 1] += g->adj_ptr[v];
  for (long i = 0; i < uniq; i++) g->adj[i] = (int)(all[i] & 0xffffffffL);
  free(all);
}

/* color_greedy, csr_matrix.tpp:2009-2363 (balance == true path used by
 * conflict_free_aposteriori with part_by_nrows_, :1501).
 *   phase 1 (:2039-2072): natural-order first-fit distance-1 colouring;
 *   phase 2 (:2099-2199): balance colours 0 and 1 only (k_color = 2), per
 *   thread, ncolors-1 steps, FIFO bins, ImbalanceTol = 0.
 * The reference runs phase 2 inside `omp parallel` on the SHARED color[]
 * (a benign race); the oracle runs the threads one after another, which is
 * one of the interleavings the reference allows.                            */
static int color_greedy(const graph_t *g, const int *weight,
                        const int *row_split, int nthreads, int *color) {
  const int V = g->V;
  int max_color = 0;
  int *mark = (int *)malloc(sizeof(int) * (size_t)(V ? V : 1));
  for (int i = 0; i < V; i++) mark[i] = 0x7fffffff;
  for (int i = 0; i < V; i++) {
    for (int p = g->adj_ptr[i]; p < g->adj_ptr[i + 1]; p++)
      mark[color[g->adj[p]]] = i;
    int j = 0;
    while (j < max_color && mark[j] == i) ++j;
    if (j == max_color) ++max_color;
    color[i] = j;
  }
  free(mark);
  const int ncolors = max_color;
  const int k_color = 2;
  if (ncolors < 2) return ncolors; /* steps = ncolors-1 = 0: nothing moves */

  int *load = (int *)malloc(sizeof(int) * (size_t)ncolors);
  char *used = (char *)malloc((size_t)ncolors);
  for (int tid = 0; tid < nthreads; tid++) {
    int row_offset = row_split[tid];
    int nrows_t = row_split[tid + 1] - row_split[tid];
    int nblk = (int)ceil(nrows_t / 16.0);
    int total_load = 0;
    memset(load, 0, sizeof(int) * (size_t)ncolors);
    for (int i = 0; i < nblk; i++) {
      int row = ((i << 4) + row_offset) >> 4;
      if (color[row] < k_color) total_load += weight[row];
      load[color[row]] += weight[row];
    }
    int mean_load = total_load / k_color;
    int *bin = (int *)malloc(sizeof(int) * (size_t)(2 * nblk + 2));
    for (int step = 0; step < ncolors - 1; ++step) {
      /* only bin[max_c] and bin[target_c] are ever touched; vertices pushed
       * to bin[target_c] are never revisited within the step, so one FIFO
       * of the max_c vertices (in thread order) is enough.                */
      int dev0 = load[0] - mean_load, dev1 = load[1] - mean_load;
      int max_c = (dev1 > dev0) ? 1 : 0; /* max_element: first maximum */
      int nb = 0;
      for (int i = 0; i < nblk; i++) {
        int row = ((i << 4) + row_offset) >> 4;
        if (color[row] == max_c) bin[nb++] = row;
      }
      int head = 0;
      while (load[max_c] - mean_load > 0 /* ImbalanceTol */ && head < nb) {
        int vid = bin[head];
        memset(used, 0, (size_t)ncolors);
        used[max_c] = 1;
        for (int p = g->adj_ptr[vid]; p < g->adj_ptr[vid + 1]; p++)
          used[color[g->adj[p]]] = 1;
        int target_c = (max_c + 1) % k_color;
        if (!used[target_c] && target_c != max_c) {
          color[vid] = target_c;
          load[max_c] -= weight[vid];
          load[target_c] += weight[vid];
        }
        head++;
      }
    }
    free(bin);
  }
  free(load);
  free(used);
  return ncolors;
}

/* =========================================================================
 * value-typed part: instantiate for double and float
 * ========================================================================= */
struct orc_sym {
  int is_f64;
  int n, nthreads, ncolors, nranges, nnz_low, nnz_diag, nvertices, nedges;
  int *row_split; /* [T+1] */
  int *color;     /* [V] (T>1) */
  struct part {
    int nrows, row_offset, nnz_low, nnz_diag, nranges;
    int *rowptr, *colind;
    void *values, *diagonal;
    int *range_ptr, *range_start, *range_end;
  } * parts;
};

#define VAL double
#define SUF(x) x##_f64
#define IS_F64 1
#include "cfs_oracle_impl.inc"
#undef VAL
#undef SUF
#undef IS_F64

#define VAL float
#define SUF(x) x##_f32
#define IS_F64 0
#include "cfs_oracle_impl.inc"
#undef VAL
#undef SUF
#undef IS_F64

void orc_sym_free(orc_sym *s) {
  if (!s) return;
  for (int t = 0; t < s->nthreads; t++) {
    struct part *p = &s->parts[t];
    free(p->rowptr);
    free(p->colind);
    free(p->values);
    free(p->diagonal);
    free(p->range_ptr);
    free(p->range_start);
    free(p->range_end);
  }
  free(s->parts);
  free(s->row_split);
  free(s->color);
  free(s);
}

/* CSRMatrix::size(), csr_matrix.tpp:189-228 (self-described FIXME formula,
 * reproduced as written for the cmp_symmetry_ case)                         */
int orc_sym_info(const orc_sym *s, orc_sym_info_t *o) {
  if (!s) return fail(-30, "null handle");
  o->nthreads = s->nthreads;
  o->ncolors = s->ncolors;
  o->nranges = s->nranges;
  o->nnz_low = s->nnz_low;
  o->nnz_diag = s->nnz_diag;
  o->nvertices = s->nvertices;
  o->nedges = s->nedges;
  size_t vs = s->is_f64 ? 8 : 4, size = 0;
  size += ((size_t)s->n + 1 * (size_t)s->nthreads) * sizeof(int);
  size += (size_t)s->nnz_low * sizeof(int);
  size += (size_t)s->nnz_low * vs;
  size += (size_t)s->nnz_diag * vs;
  if (s->nthreads > 1) {
    size += ((size_t)s->ncolors + 1) * sizeof(int);
    size += 2 * (size_t)s->nranges * sizeof(int);
  }
  o->size_bytes = size;
  return 0;
}

int orc_sym_colors(const orc_sym *s, int *color_out) {
  if (!s || !s->color) return fail(-31, "no colouring (T == 1)");
  memcpy(color_out, s->color, sizeof(int) * (size_t)s->nvertices);
  return 0;
}

int orc_sym_partition(const orc_sym *s, int t, int *row_offset, int *nrows,
                      const int **rowptr, const int **colind,
                      const void **values, const void **diagonal) {
  if (!s || t < 0 || t >= s->nthreads) return fail(-32, "bad partition");
  const struct part *p = &s->parts[t];
  *row_offset = p->row_offset;
  *nrows = p->nrows;
  *rowptr = p->rowptr;
  *colind = p->colind;
  *values = p->values;
  *diagonal = p->diagonal;
  return 0;
}
