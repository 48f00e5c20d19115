/*
 * cfs_oracle.h -- CPU oracle for the symmetric-SpMV hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 *
 * It is a plain-C restatement of the reference algorithm (athelaf/cfs-spmv);
 * every function cites the reference file:line it follows (paths relative to
 * the reference root).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - MMF reader + CSR construction: PINNED against the genuine reference
 *     reader (src/mmf.cpp + include/io/mmf.hpp compile on their own and are
 *     built into oracle/_ref/ by oracle/Makefile; tests/golden holds the
 *     CSR it produced for the committed .mtx fixtures).
 *   - Floating-point kernels (csr_matrix.tpp): PARITY UNPINNED against
 *     reference outputs.  include/matrix/csr_matrix.hpp needs the TBB headers
 *     and an autoheader-generated config.h, neither of which exists in this
 *     image, so the numeric path of the reference is unbuildable here and the
 *     reference ships no golden vectors.  The restatement is checked against
 *     the reference's own acceptance criterion (SSS result vs plain CSR result,
 *     test/test_spmv_mmf.cpp:85-109) and against an exactly-rounded
 *     long-double arbiter.
 */
#ifndef CFS_ORACLE_H
#define CFS_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error string of the last failing call on this thread ("" if none) */
const char *orc_last_error(void);
void orc_free(void *p);

/* ---- a1/a2: Matrix-Market reader + CSR constructor ------------------------
 * include/io/mmf.hpp:179-343, src/mmf.cpp:6-44, include/matrix/csr_matrix.tpp:8-111
 * Returns 0 or a negative code (the reference prints and exit(1)s instead).
 * `symmetric` = the file header says "symmetric".  The arrays are the FULL
 * (expanded) CSR exactly as CSRMatrix holds it before tune().            */
int orc_mmf_load_f64(const char *path, int *nrows, int *ncols, int *nnz,
                     int *symmetric, int **rowptr, int **colind,
                     double **values);
int orc_mmf_load_f32(const char *path, int *nrows, int *ncols, int *nnz,
                     int *symmetric, int **rowptr, int **colind,
                     float **values);

/* ---- a6: partition_by_nrows (csr_matrix.tpp:403-435) ---------------------- */
int orc_partition_by_nrows(int nrows, int nthreads, int *row_split /*[T+1]*/);
/* partition_by_nnz, unsymmetric branch (csr_matrix.tpp:437-541) */
int orc_partition_by_nnz(int nrows, const int *rowptr, int nthreads,
                         int *row_split /*[T+1]*/);

/* ---- a7..a11: symmetric (SSS) path ----------------------------------------
 * build = tune() for a symmetric matrix (csr_matrix.tpp:230-310):
 *   T == 1 -> serial() split (:641-706) + cpu_mv_sym_serial (:2706-2729)
 *   T  > 1 -> partition_by_nrows + conflict_free_aposteriori (:1204-1639)
 *             + color_greedy (:2009-2363) + cpu_mv_sym_conflict_free_v2
 *             (:2965-3028, _USE_BARRIER build)                              */
typedef struct orc_sym orc_sym;

orc_sym *orc_sym_build_f64(int n, const int *rowptr, const int *colind,
                           const double *values, int nthreads);
orc_sym *orc_sym_build_f32(int n, const int *rowptr, const int *colind,
                           const float *values, int nthreads);
void orc_sym_spmv_f64(const orc_sym *s, double *y, const double *x);
void orc_sym_spmv_f32(const orc_sym *s, float *y, const float *x);
void orc_sym_free(orc_sym *s);

typedef struct {
  int nthreads, ncolors, nranges, nnz_low, nnz_diag, nvertices, nedges;
  size_t size_bytes; /* CSRMatrix::size() formula, csr_matrix.tpp:189-228 */
} orc_sym_info_t;
int orc_sym_info(const orc_sym *s, orc_sym_info_t *out);
/* colour of every 16-row block (length ceil(n/16)); T>1 only */
int orc_sym_colors(const orc_sym *s, int *color_out);
/* lower-triangle CSR of partition t (pointers stay owned by s) */
int orc_sym_partition(const orc_sym *s, int t, int *row_offset, int *nrows,
                      const int **rowptr, const int **colind,
                      const void **values, const void **diagonal);

/* ---- a12: plain CSR kernels (csr_matrix.tpp:2664-2704) -------------------- */
void orc_csr_spmv_f64(int n, const int *rowptr, const int *colind,
                      const double *values, int nthreads, const int *row_split,
                      double *y, const double *x);
void orc_csr_spmv_f32(int n, const int *rowptr, const int *colind,
                      const float *values, int nthreads, const int *row_split,
                      float *y, const float *x);

/* ---- arbiter: y = A x accumulated in long double, plus sum |a_ij||x_j| ----
 * (not in the reference; used to arbitrate 1e-12 claims between two
 * differently-ordered fp64 sums, SURVEY.md section 7 "hard parts")          */
void orc_csr_spmv_ld_f64(int n, const int *rowptr, const int *colind,
                         const double *values, const double *x, double *y,
                         double *absrow);
void orc_csr_spmv_ld_f32(int n, const int *rowptr, const int *colind,
                         const float *values, const float *x, double *y,
                         double *absrow);

/* reference pass criterion, include/utils/platform.hpp:27-37 */
int orc_is_equal_f64(double x, double y);
int orc_is_equal_f32(float x, float y);

/* CFS_NUM_THREADS semantics, src/runtime.cpp:10-21 */
int orc_get_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
