/*
 * cfs_synth.c -- structure-faithful synthetic stand-ins for the SuiteSparse
 * matrices BASELINE.json names (the real .mtx files are not available offline).
 *
 * Workload generation only: no SpMV arithmetic lives here.  Definitions follow
 * symmetric, full nonzero diagonal, int32 indices,
 * seed = FNV-1a(name), off-diagonal values U(-1,0), diag = 1 + sum |row|,
 * x_i = 0.01 + 0.41 u_i with u from a 64-bit LCG seeded 42 (mirrors the
 * U(0.01,0.42) of the reference driver, bench/bench_spmv_mmf.cpp:125).
 *
 * The generator emits the FULL (both triangles) CSR exactly as the reference's
 * CSRMatrix holds it before tune() (rows ascending, columns ascending inside a
 * row, 0-based; include/matrix/csr_matrix.tpp:74-107), because that is what
 * the C ABI (include/cfs_hip.h) takes.
 */
#define _GNU_SOURCE
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t splitmix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ULL;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
static inline double u01(uint64_t h) { /* (0,1] */
  return ((double)(h >> 11) + 1.0) * (1.0 / 9007199254740992.0);
}
static uint64_t fnv1a(const char *s) {
  uint64_t h = 0xcbf29ce484222325ULL;
  for (; *s; s++) {
    h ^= (unsigned char)*s;
    h *= 0x100000001b3ULL;
  }
  return h;
}
/* symmetric off-diagonal value in [-1, 0): a function of the unordered pair */
static inline double offdiag(uint64_t seed, int i, int j) {
  uint64_t lo = (uint64_t)(i < j ? i : j), hi = (uint64_t)(i < j ? j : i);
  return -u01(splitmix64(seed ^ (hi << 32 | lo)));
}

typedef struct {
  int kind;           /* 0 = grid stencil, 1 = banded-random, 2 = point cloud (unstructured) */
  int n;              /* rows */
  int nx, ny, nz, dof;/* grid */
  int pts;            /* 27 or 7 */
  int irregular;      /* ldoor-like: 2% fat rows, 5% thin rows */
  int per_row, mean_off, cap_off; /* banded-random */
  uint64_t seed;
  /* point cloud: lower node neighbours (ascending) of every node, in the final numbering */
  int knn, bfs;
  long *adj_ptr;
  int *adj;
  /* tetmesh (kind 2 with var_dof): 1..4 unknowns per node, node_off = first row of a node;
   * a row couples with ~3/4 of a neighbour node's unknowns, each row its own choice */
  int var_dof, nnodes;
  int *node_off;
} spec_t;

/* strictly-lower column list of row i (ascending, unique).  Returns count. */
static int lower_cols(const spec_t *sp, int i, int *out, int cap) {
  int cnt = 0;
  if (sp->kind == 2 && sp->var_dof) {
    /* tetrahedral-mesh-like, unequal unknowns per node: the rows of one node do NOT repeat
     * each other's columns (a row keeps an entry of a neighbour node's block with
     * probability 3/4, decided by the unordered pair of row indices: symmetric) */
    int lo = 0, hi = sp->nnodes; /* node of row i: last node_off <= i */
    while (hi - lo > 1) {
      const int m = (lo + hi) >> 1;
      if (sp->node_off[m] <= i) lo = m;
      else hi = m;
    }
    const int node = lo;
    for (long k = sp->adj_ptr[node]; k < sp->adj_ptr[node + 1]; k++) {
      const int u = sp->adj[k];
      for (int c = sp->node_off[u]; c < sp->node_off[u + 1]; c++)
        if (c < i && cnt < cap && (splitmix64(sp->seed ^ 0x7e7a11ULL ^ ((uint64_t)i << 32 | (uint64_t)c)) & 3) != 0)
          out[cnt++] = c;
    }
    for (int c = sp->node_off[node]; c < i; c++)
      if (cnt < cap) out[cnt++] = c; /* the node's own block is dense */
    return cnt; /* ascending (neighbours ascend, rows of a node are consecutive) and unique */
  }
  if (sp->kind == 3) {
    /* power-law graph: a heavy-tailed number of lower entries per row (a few rows with
     * thousands: hubs), half of them inside a window of 2 000 rows (communities), half
     * preferential -- concentrated on low indices, which become hub COLUMNS (long tails) */
    const uint64_t h0 = splitmix64(sp->seed ^ ((uint64_t)i * 0x9e3779b97f4a7c15ULL));
    int k = 8 + (int)(12.0 * (pow(u01(h0), -0.6) - 1.0));
    if (k > 2000) k = 2000;
    if (k > i) k = i;
    for (int q = 0; q < k && cnt < cap; q++) {
      const uint64_t h = splitmix64(h0 + (uint64_t)(q + 1) * 0xd6e8feb86659fd93ULL);
      int c;
      if (q & 1) {
        const int win = i < 2000 ? i : 2000;
        c = i - 1 - (int)((h >> 8) % (uint64_t)win);
      } else {
        const double v = u01(h);
        c = (int)((double)i * v * v * v);
      }
      if (c >= 0 && c < i) out[cnt++] = c;
    }
    goto sort_unique;
  }
  if (sp->kind == 2) {
    /* unstructured: dof x dof blocks with the lower node neighbours, then the own node */
    const int dof = sp->dof, node = i / dof;
    for (long k = sp->adj_ptr[node]; k < sp->adj_ptr[node + 1]; k++)
      for (int e = 0; e < dof; e++) {
        const int c = sp->adj[k] * dof + e;
        if (c < sp->n && cnt < cap) out[cnt++] = c;
      }
    for (int e = 0; e < i % dof; e++)
      if (cnt < cap) out[cnt++] = node * dof + e;
    return cnt; /* ascending and unique by construction */
  }
  if (sp->kind == 1) {
    /* pdb1HYS-like: per_row draws at offsets 1 + floor(Exp(mean)) capped */
    for (int k = 0; k < sp->per_row && cnt < cap; k++) {
      uint64_t h = splitmix64(sp->seed ^ ((uint64_t)i * 1315423911ULL + (uint64_t)k));
      double e = -log(u01(h)) * sp->mean_off;
      int off = 1 + (int)e;
      if (off > sp->cap_off) off = sp->cap_off;
      int c = i - off;
      if (c >= 0) out[cnt++] = c;
    }
  } else {
    int dof = sp->dof, node = i / dof, d = i % dof;
    int x = node % sp->nx, y = (node / sp->nx) % sp->ny, z = node / (sp->nx * sp->ny);
    for (int dz = -1; dz <= 0; dz++)
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          if (sp->pts == 7 && (abs(dx) + abs(dy) + abs(dz)) > 1) continue;
          int X = x + dx, Y = y + dy, Z = z + dz;
          if (X < 0 || X >= sp->nx || Y < 0 || Y >= sp->ny || Z < 0) continue;
          int nb = X + sp->nx * (Y + sp->ny * Z);
          if (nb > node) continue;
          for (int e = 0; e < dof; e++) {
            int c = nb * dof + e;
            if (c < i && cnt < cap) out[cnt++] = c;
          }
        }
    (void)d;
    if (sp->irregular) {
      uint64_t h = splitmix64(sp->seed ^ (0xabcdef12345ULL + (uint64_t)i));
      double u = u01(h);
      if (u < 0.02) { /* fat row: ~3x extra lower entries in a local window */
        int extra = 3 * cnt, win = 40000;
        for (int k = 0; k < extra && cnt < cap; k++) {
          uint64_t h2 = splitmix64(h + (uint64_t)k * 0x9e3779b97f4a7c15ULL);
          int c = i - 1 - (int)(h2 % (uint64_t)win);
          if (c >= 0) out[cnt++] = c;
        }
      } else if (u < 0.07) { /* thin row: keep only the 4 nearest */
        /* applied after sort below */
      }
    }
  }
sort_unique:
  /* sort + unique */
  for (int a = 1; a < cnt; a++) {
    int v = out[a], b = a - 1;
    while (b >= 0 && out[b] > v) {
      out[b + 1] = out[b];
      b--;
    }
    out[b + 1] = v;
  }
  int u = 0;
  for (int a = 0; a < cnt; a++)
    if (a == 0 || out[a] != out[a - 1]) out[u++] = out[a];
  cnt = u;
  if (sp->kind == 0 && sp->irregular) {
    uint64_t h = splitmix64(sp->seed ^ (0xabcdef12345ULL + (uint64_t)i));
    double uu = u01(h);
    if (uu >= 0.02 && uu < 0.07 && cnt > 4) {
      memmove(out, out + (cnt - 4), 4 * sizeof(int));
      cnt = 4;
    }
  }
  return cnt;
}

static int make_spec(const char *name, double scale, spec_t *sp) {
  memset(sp, 0, sizeof *sp);
  sp->seed = fnv1a(name);
  if (scale <= 0) scale = 1.0;
  double s3 = cbrt(scale);
  if (!strcmp(name, "pdb1HYS")) {
    sp->kind = 1;
    sp->n = (int)(36417 * scale);
    sp->per_row = 62; /* ~59 unique lower nz/row after dedup */
    sp->mean_off = 300;
    sp->cap_off = 4000;
  } else if (!strcmp(name, "pwtk")) {
    sp->n = (int)(217918 * scale);
    sp->dof = 2; sp->pts = 27;
    sp->nx = (int)ceil(48 * s3); sp->ny = (int)ceil(48 * s3); sp->nz = (int)ceil(48 * s3);
  } else if (!strcmp(name, "ldoor") || !strcmp(name, "ldoor_regular")) {
    /* "ldoor_regular": the same 7-point x 7-dof grid without the fat / thin rows (what is
     * left of the ldoor stand-in once its single-use entries are far entries) */
    sp->n = (int)(952203 * scale);
    sp->dof = 7; sp->pts = 7; sp->irregular = strcmp(name, "ldoor") == 0;
    sp->nx = (int)ceil(52 * s3); sp->ny = (int)ceil(52 * s3); sp->nz = (int)ceil(51 * s3);
  } else if (!strcmp(name, "Flan_1565")) {
    sp->n = (int)(1564794 * scale);
    sp->dof = 3; sp->pts = 27;
    sp->nx = (int)ceil(81 * s3); sp->ny = (int)ceil(81 * s3); sp->nz = (int)ceil(80 * s3);
  } else if (!strcmp(name, "Queen_4147")) {
    sp->n = (int)(4147110 * scale);
    sp->dof = 3; sp->pts = 27;
    sp->nx = (int)ceil(112 * s3); sp->ny = (int)ceil(112 * s3); sp->nz = (int)ceil(111 * s3);
  } else if (!strcmp(name, "unstruct") || !strcmp(name, "unstruct_bfs")) {
    /* a non-regular stand-in of Flan_1565's size: random points in the unit cube, each
     * node tied to its 21 nearest neighbours (symmetrised: ~24 neighbours, ~75 nonzeros
     * per row with 3 dof, as the real hex-mesh matrix), 3 dof per node.  "unstruct"
     * keeps the random node numbering (no locality at all in the natural order);
     * "unstruct_bfs" renumbers breadth-first from a corner (an RCM-like band). */
    sp->kind = 2;
    sp->n = (int)(1564794 * scale);
    sp->dof = 3;
    sp->knn = 21;
    sp->bfs = !strcmp(name, "unstruct_bfs");
    sp->seed = fnv1a("unstruct"); /* both orders are the SAME graph */
  } else if (!strcmp(name, "tetmesh")) {
    /* tetrahedral-mesh-like: the same kind of point-cloud graph (21 nearest neighbours,
     * breadth-first numbering), but 1..4 unknowns per node (mixed pressure / velocity /
     * temperature) and rows of a node that are NOT prefixes of one another: the shapes for
     * which the slot leaders and sibling chains of the tile format pay least */
    sp->kind = 2;
    sp->var_dof = 1;
    sp->dof = 3; /* nominal: nodes = n / 2.8 */
    sp->n = (int)(1500000 * scale);
    sp->knn = 21;
    sp->bfs = 1;
  } else if (!strcmp(name, "powerlaw")) {
    /* power-law-degree symmetric graph: hub rows and long tails */
    sp->kind = 3;
    sp->n = (int)(1000000 * scale);
  } else {
    return -1;
  }
  if (sp->n < 1) sp->n = 1;
  if (sp->kind == 0) {
    /* the grid must hold ceil(n/dof) nodes; grow nz if rounding fell short */
    long need = ((long)sp->n + sp->dof - 1) / sp->dof;
    while ((long)sp->nx * sp->ny * sp->nz < need) sp->nz++;
  }
  return 0;
}


/* ---- point cloud -> k-nearest-neighbour graph (kind 2) ------------------------------ */
static int cmp_u64(const void *a, const void *b) {
  const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : x > y;
}
static int build_cloud(spec_t *sp) {
  /* (var_dof: 1..4 unknowns per node, 2.8 on average) */
  const int N = sp->var_dof ? (int)((double)sp->n / 2.8) + 1 : (sp->n + sp->dof - 1) / sp->dof, K = sp->knn;
  float *px = (float *)malloc(sizeof(float) * 3 * (size_t)N);
  if (!px) return -1;
  for (int v = 0; v < N; v++)
    for (int d = 0; d < 3; d++)
      px[3 * (size_t)v + d] = (float)u01(splitmix64(sp->seed ^ ((uint64_t)v * 3 + (uint64_t)d + 0x5bd1e995ULL)));
  /* buckets of ~6 points */
  int G = (int)ceil(cbrt((double)N / 6.0));
  if (G < 1) G = 1;
  const long ncell = (long)G * G * G;
  int *cstart = (int *)calloc((size_t)ncell + 1, sizeof(int));
  int *cell = (int *)malloc(sizeof(int) * (size_t)N), *order = (int *)malloc(sizeof(int) * (size_t)N);
  for (int v = 0; v < N; v++) {
    int cx = (int)(px[3 * (size_t)v] * G), cy = (int)(px[3 * (size_t)v + 1] * G), cz = (int)(px[3 * (size_t)v + 2] * G);
    if (cx >= G) cx = G - 1;
    if (cy >= G) cy = G - 1;
    if (cz >= G) cz = G - 1;
    cell[v] = cx + G * (cy + G * cz);
    cstart[cell[v] + 1]++;
  }
  for (long c = 0; c < ncell; c++) cstart[c + 1] += cstart[c];
  {
    int *fill = (int *)malloc(sizeof(int) * (size_t)ncell);
    memcpy(fill, cstart, sizeof(int) * (size_t)ncell);
    for (int v = 0; v < N; v++) order[fill[cell[v]]++] = v;
    free(fill);
  }
  /* K nearest of every node among the points of the 27 (or, where too few, 125) cells around it */
  uint64_t *edges = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)N * (size_t)K);
  long nedge_cap = (long)N * K;
#pragma omp parallel for schedule(dynamic, 256)
  for (int v = 0; v < N; v++) {
    int best[64];
    float bd[64];
    int nb = 0;
    const int cx = cell[v] % G, cy = (cell[v] / G) % G, cz = cell[v] / (G * G);
    for (int R = 1; R <= 3; R++) {
      nb = 0;
      for (int dz = -R; dz <= R; dz++)
        for (int dy = -R; dy <= R; dy++)
          for (int dx = -R; dx <= R; dx++) {
            const int X = cx + dx, Y = cy + dy, Z = cz + dz;
            if (X < 0 || Y < 0 || Z < 0 || X >= G || Y >= G || Z >= G) continue;
            const long c = X + (long)G * (Y + (long)G * Z);
            for (int q = cstart[c]; q < cstart[c + 1]; q++) {
              const int u = order[q];
              if (u == v) continue;
              const float ax = px[3 * (size_t)u] - px[3 * (size_t)v], ay = px[3 * (size_t)u + 1] - px[3 * (size_t)v + 1],
                          az = px[3 * (size_t)u + 2] - px[3 * (size_t)v + 2];
              const float dd = ax * ax + ay * ay + az * az;
              if (nb < K || dd < bd[nb - 1]) { /* insertion into the sorted K best */
                int p = nb < K ? nb++ : K - 1;
                while (p > 0 && (bd[p - 1] > dd || (bd[p - 1] == dd && best[p - 1] > u))) {
                  bd[p] = bd[p - 1];
                  best[p] = best[p - 1];
                  p--;
                }
                bd[p] = dd;
                best[p] = u;
              }
            }
          }
      if (nb >= K || R * 2 + 1 >= G) break;
    }
    for (int k = 0; k < K; k++) {
      uint64_t key = ~0ULL; /* unused slots sort to the end */
      if (k < nb) {
        const uint64_t a = (uint64_t)(v < best[k] ? v : best[k]), b = (uint64_t)(v < best[k] ? best[k] : v);
        key = (b << 32) | a; /* (higher, lower) */
      }
      edges[(size_t)v * K + k] = key;
    }
  }
  qsort(edges, (size_t)nedge_cap, sizeof(uint64_t), cmp_u64);
  long ne = 0;
  for (long k = 0; k < nedge_cap; k++)
    if (edges[k] != ~0ULL && (ne == 0 || edges[k] != edges[ne - 1])) edges[ne++] = edges[k];
  /* numbering: as generated (random), or breadth-first from the node nearest the origin */
  int *label = (int *)malloc(sizeof(int) * (size_t)N);
  for (int v = 0; v < N; v++) label[v] = v;
  if (sp->bfs) {
    long *ap = (long *)calloc((size_t)N + 1, sizeof(long));
    for (long k = 0; k < ne; k++) {
      ap[(edges[k] >> 32) + 1]++;
      ap[(edges[k] & 0xffffffffULL) + 1]++;
    }
    for (int v = 0; v < N; v++) ap[v + 1] += ap[v];
    int *aj = (int *)malloc(sizeof(int) * (size_t)(2 * ne + 1));
    long *fill = (long *)malloc(sizeof(long) * (size_t)N);
    memcpy(fill, ap, sizeof(long) * (size_t)N);
    for (long k = 0; k < ne; k++) {
      const int hi = (int)(edges[k] >> 32), lo = (int)(edges[k] & 0xffffffffULL);
      aj[fill[hi]++] = lo;
      aj[fill[lo]++] = hi;
    }
    free(fill);
    int start = 0;
    float bestd = 1e30f;
    for (int v = 0; v < N; v++) {
      const float dd = px[3 * (size_t)v] + px[3 * (size_t)v + 1] + px[3 * (size_t)v + 2];
      if (dd < bestd) {
        bestd = dd;
        start = v;
      }
    }
    int *queue = (int *)malloc(sizeof(int) * (size_t)N);
    for (int v = 0; v < N; v++) label[v] = -1;
    int head = 0, tail = 0, next_seed = 0;
    queue[tail++] = start;
    label[start] = 0;
    int cnt = 1;
    while (head < N) {
      if (head == tail) { /* disconnected remainder */
        while (label[next_seed] >= 0) next_seed++;
        label[next_seed] = cnt++;
        queue[tail++] = next_seed;
      }
      const int v = queue[head++];
      for (long q = ap[v]; q < ap[v + 1]; q++) /* neighbours in stored (ascending-id) order */
        if (label[aj[q]] < 0) {
          label[aj[q]] = cnt++;
          queue[tail++] = aj[q];
        }
    }
    free(queue);
    free(ap);
    free(aj);
  }
  /* lower node lists in the final numbering */
  for (long k = 0; k < ne; k++) {
    const uint64_t a = (uint64_t)label[edges[k] >> 32], b = (uint64_t)label[edges[k] & 0xffffffffULL];
    edges[k] = ((a > b ? a : b) << 32) | (a > b ? b : a);
  }
  qsort(edges, (size_t)ne, sizeof(uint64_t), cmp_u64);
  sp->adj_ptr = (long *)calloc((size_t)N + 1, sizeof(long));
  sp->adj = (int *)malloc(sizeof(int) * (size_t)(ne + 1));
  for (long k = 0; k < ne; k++) {
    sp->adj_ptr[(edges[k] >> 32) + 1]++;
    sp->adj[k] = (int)(edges[k] & 0xffffffffULL); /* sorted by (higher, lower): in place */
  }
  for (int v = 0; v < N; v++) sp->adj_ptr[v + 1] += sp->adj_ptr[v];
  if (sp->var_dof) { /* unknowns per node: 20 % 1, 10 % 2, 40 % 3, 30 % 4; n = their sum */
    sp->nnodes = N;
    sp->node_off = (int *)malloc(sizeof(int) * ((size_t)N + 1));
    sp->node_off[0] = 0;
    for (int v = 0; v < N; v++) {
      const unsigned r = (unsigned)(splitmix64(sp->seed ^ 0xd0fULL ^ (uint64_t)v) % 10u);
      const int d = r < 2 ? 1 : (r < 3 ? 2 : (r < 7 ? 3 : 4));
      sp->node_off[v + 1] = sp->node_off[v] + d;
    }
    sp->n = sp->node_off[N];
  }
  free(edges);
  free(label);
  free(px);
  free(cstart);
  free(cell);
  free(order);
  return 0;
}

/* Generate matrix `name` ("pdb1HYS", "pwtk", "ldoor", "Flan_1565",
 * "Queen_4147"), optionally scaled down (scale in (0,1]: n and the grid shrink
 * together so the structure is preserved).  Output: full CSR, values as fp64
 * (callers round to fp32 for the single-precision build).  Arrays are
 * malloc'ed; free with cfs_synth_free.  Returns 0, or -1 for an unknown name. */
int cfs_synth_generate(const char *name, double scale, int *n_out,
                       long *nnz_full_out, long *nnz_low_out, int **rowptr_out,
                       int **colind_out, double **values_out) {
  spec_t sp;
  if (make_spec(name, scale, &sp) != 0) return -1;
  const int CAP = 2048;
  if (sp.kind == 2 && build_cloud(&sp) != 0) return -3;
  const int n = sp.n; /* (tetmesh: known once the unknowns per node are drawn) */
  /* pass 1: lower counts */
  long *lptr = (long *)calloc((size_t)n + 1, sizeof(long));
#pragma omp parallel
  {
    int *buf = (int *)malloc(sizeof(int) * CAP);
#pragma omp for schedule(dynamic, 1024)
    for (int i = 0; i < n; i++) lptr[i + 1] = lower_cols(&sp, i, buf, CAP);
    free(buf);
  }
  for (int i = 0; i < n; i++) lptr[i + 1] += lptr[i];
  long nnz_low = lptr[n];
  int *lcol = (int *)malloc(sizeof(int) * (size_t)(nnz_low ? nnz_low : 1));
#pragma omp parallel
  {
    int *buf = (int *)malloc(sizeof(int) * CAP);
#pragma omp for schedule(dynamic, 1024)
    for (int i = 0; i < n; i++) {
      int c = lower_cols(&sp, i, buf, CAP);
      memcpy(lcol + lptr[i], buf, sizeof(int) * (size_t)c);
    }
    free(buf);
  }
  /* full pattern: row i = lower(i) + diag + upper(i), upper(i) = rows r > i
   * having i in lower(r): count by transposition                            */
  long nnz_full = 2 * nnz_low + n;
  if (nnz_full > 2147483647L) {
    free(lptr);
    free(lcol);
    return -2; /* int32 nnz is API (src/csr.cpp:10-11) */
  }
  int *rowptr = (int *)malloc(sizeof(int) * ((size_t)n + 1));
  int *ucnt = (int *)calloc((size_t)n, sizeof(int));
  for (long k = 0; k < nnz_low; k++) ucnt[lcol[k]]++;
  rowptr[0] = 0;
  for (int i = 0; i < n; i++)
    rowptr[i + 1] = rowptr[i] + (int)(lptr[i + 1] - lptr[i]) + 1 + ucnt[i];
  int *colind = (int *)malloc(sizeof(int) * (size_t)nnz_full);
  double *values = (double *)malloc(sizeof(double) * (size_t)nnz_full);
  int *upos = (int *)malloc(sizeof(int) * (size_t)n);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    int p = rowptr[i];
    for (long k = lptr[i]; k < lptr[i + 1]; k++) {
      colind[p] = lcol[k];
      values[p] = offdiag(sp.seed, i, lcol[k]);
      p++;
    }
    colind[p] = i; /* diagonal placeholder, value set below */
    values[p] = 0.0;
    upos[i] = p + 1;
  }
  /* upper entries: rows visited ascending => columns ascend inside each row */
  for (int r = 0; r < n; r++)
    for (long k = lptr[r]; k < lptr[r + 1]; k++) {
      int i = lcol[k];
      int p = upos[i]++;
      colind[p] = r;
      values[p] = offdiag(sp.seed, i, r);
    }
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    double s = 0.0;
    int pd = -1;
    for (int p = rowptr[i]; p < rowptr[i + 1]; p++) {
      if (colind[p] == i) pd = p;
      else s += fabs(values[p]);
    }
    values[pd] = 1.0 + s;
  }
  free(lptr);
  free(lcol);
  free(ucnt);
  free(upos);
  free(sp.adj_ptr);
  free(sp.adj);
  free(sp.node_off);
  *n_out = n;
  *nnz_full_out = nnz_full;
  *nnz_low_out = nnz_low;
  *rowptr_out = rowptr;
  *colind_out = colind;
  *values_out = values;
  return 0;
}

void cfs_synth_free(void *p) { free(p); }

/* x_i = 0.01 + 0.41 u_i, u from a 64-bit LCG (Knuth MMIX constants) seeded
 * `seed` (42 for the benchmark vector).                                      */
void cfs_synth_x(int n, uint64_t seed, double *x) {
  uint64_t s = seed;
  for (int i = 0; i < n; i++) {
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    x[i] = 0.01 + 0.41 * ((double)(s >> 11) * (1.0 / 9007199254740992.0));
  }
}

/* write a symmetric Matrix-Market file (lower triangle + diagonal, 1-based,
 * single-space separated, trailing newline -- the dialect the reference reader
 * accepts, src/mmf.cpp:26-44) from a full CSR.                               */
int cfs_synth_write_mtx(const char *path, int n, const int *rowptr,
                        const int *colind, const double *values, int general) {
  FILE *f = fopen(path, "w");
  if (!f) return -1;
  long cnt = 0;
  for (int i = 0; i < n; i++)
    for (int p = rowptr[i]; p < rowptr[i + 1]; p++)
      if (general || colind[p] <= i) cnt++;
  fprintf(f, "%%%%MatrixMarket matrix coordinate real %s\n",
          general ? "general" : "symmetric");
  fprintf(f, "%% generated by cfs_synth.c\n");
  fprintf(f, "%d %d %ld\n", n, n, cnt);
  for (int i = 0; i < n; i++)
    for (int p = rowptr[i]; p < rowptr[i + 1]; p++)
      if (general || colind[p] <= i)
        fprintf(f, "%d %d %.17g\n", i + 1, colind[p] + 1, values[p]);
  fclose(f);
  return 0;
}
