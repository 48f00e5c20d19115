// cfs.cpp -- SparseMatrix factory and SpDMV functor (reference:
// include/matrix/sparse_matrix.tpp:13-24, include/kernel/sparse_kernel.tpp:8-27;
// instantiated for <int,float> and <int,double> as in src/cfs.cpp:11-21).
#include <cassert>
#include <iostream>

#include "cfs.hpp"

namespace cfs {
namespace matrix {
namespace sparse {

template <typename IndexT, typename ValueT>
SparseMatrix<IndexT, ValueT> *SparseMatrix<IndexT, ValueT>::create(const std::string &filename,
                                                                   Format format,
                                                                   Platform platform) {
  switch (format) {
  case Format::sss: return new CSRMatrix<IndexT, ValueT>(filename, platform, true);
  case Format::hyb: return new CSRMatrix<IndexT, ValueT>(filename, platform, true, true);
  default: return new CSRMatrix<IndexT, ValueT>(filename, platform);
  }
}

template class SparseMatrix<int, float>;
template class SparseMatrix<int, double>;

} // namespace sparse
} // namespace matrix

namespace kernel {
namespace sparse {

template <typename IndexType, typename ValueType>
SpDMV<IndexType, ValueType>::SpDMV(SparseMatrix<IndexType, ValueType> *A, Tuning t) : A_(A) {
  if (A_->tune(Kernel::SpDMV, t)) {
#ifdef _LOG_INFO
    std::cout << "[INFO]: matrix format was tuned successfully" << std::endl;
#endif
  }
}

template <typename IndexType, typename ValueType>
void SpDMV<IndexType, ValueType>::operator()(ValueType *__restrict y, const int M,
                                             const ValueType *__restrict x, const int N) {
  assert(A_->nrows() == M);
  assert(A_->ncols() == N);
  A_->dense_vector_multiply(y, x);
}

template struct SpDMV<int, float>;
template struct SpDMV<int, double>;

} // namespace sparse
} // namespace kernel
} // namespace cfs
