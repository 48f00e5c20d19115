// kernel/sparse_kernel.hpp -- the SpDMV functor (reference:
// include/kernel/sparse_kernel.hpp:17-27, sparse_kernel.tpp:9-27): the
// constructor tunes the matrix, the call operator checks the dimensions and
// multiplies.
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
public:
  SpDMV() = delete;
  SpDMV(SparseMatrix<IndexType, ValueType> *A, Tuning t = Tuning::Aggressive);
  void operator()(ValueType *__restrict y, const int M, const ValueType *__restrict x,
                  const int N);

private:
  SparseMatrix<IndexType, ValueType> *A_;
};

} // namespace sparse
} // namespace kernel
} // namespace cfs

#endif
