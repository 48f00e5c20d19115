// kernel/sparse_kernel.hpp -- SpDMV, the sparse-matrix times dense-vector functor.
//
// Mirrors the reference's operator (include/kernel/sparse_kernel.hpp:17-27,
// sparse_kernel.tpp:9-27) so that callers compile unchanged:
//
//     SpDMV<int, double> spmv(A);        // tunes A: builds the MI355X tile schedule
//     spmv(y, M, x, N);                  // y <- A x ; M == A->nrows(), N == A->ncols()
//
// x and y may be host pointers (staged through HBM on every call -- the
// unmodified-caller path) or Platform::gpu pointers from internal_alloc (resident:
// the call is enqueued; cfs::util::runtime::synchronize() or internal_copy wait).
// y is fully overwritten and needs no zeroing.  The functor does not own A.
#ifndef CFS_SPARSE_KERNEL_HPP
#define CFS_SPARSE_KERNEL_HPP

#include <cassert>

#include "cfs_config.hpp"
#include "matrix/sparse_matrix.hpp"

namespace cfs {

using namespace matrix::sparse;

namespace kernel {
namespace sparse {

template <typename IndexType, typename ValueType> struct SpDMV {
  typedef SparseMatrix<IndexType, ValueType> matrix_type;

  // tune() happens here, as in the reference (sparse_kernel.tpp:9-13); what the
  // Tuning levels mean on the GPU is described at CSRMatrix::tune (src/csr.cpp)
  SpDMV(matrix_type *A, Tuning t = Tuning::Aggressive);
  SpDMV() = delete;

  // dimensions are asserted, then forwarded to A->dense_vector_multiply(y, x)
  void operator()(ValueType *__restrict y, const int M, const ValueType *__restrict x,
                  const int N);

  // the operand this functor was built for (not in the reference; read-only helper)
  const matrix_type *matrix() const { return A_; }

private:
  matrix_type *A_;
};

} // namespace sparse
} // namespace kernel
} // namespace cfs

#endif
