// matrix/sparse_matrix.hpp -- abstract sparse matrix + factory, the L3 boundary
// callers use (reference: include/matrix/sparse_matrix.hpp:23-41,
// include/matrix/sparse_matrix.tpp:13-24).  Same names and signatures; the
// default Platform is gpu in this build, so unmodified callers
// (`create(file, Format::sss)`) land on the MI355X path.
#ifndef CFS_SPARSE_MATRIX_HPP
#define CFS_SPARSE_MATRIX_HPP

#include <cstddef>
#include <string>

#include "cfs_config.hpp"
#include "utils/platform.hpp"

namespace cfs {

using namespace util;

namespace matrix {
namespace sparse {

template <typename IndexT, typename ValueT> class SparseMatrix {
public:
  virtual ~SparseMatrix() {}
  virtual int nrows() const = 0;
  virtual int ncols() const = 0;
  virtual int nnz() const = 0; // expanded (both triangles) count, as in the reference
  virtual bool symmetric() const = 0;
  virtual size_t size() const = 0;
  virtual Platform platform() const = 0;
  virtual bool tune(Kernel k, Tuning t = Tuning::Aggressive) = 0;
  // y <- A x.  y is fully overwritten; x, y may be host pointers (staged, slow)
  // or Platform::gpu pointers from internal_alloc (resident, fast).
  virtual void dense_vector_multiply(ValueT *__restrict y, const ValueT *__restrict x) = 0;

  // Matrix-Market file -> matrix.  Format::sss keeps only the lower triangle
  // when the file is symmetric and silently falls back to CSR when it is not;
  // Format::hyb is accepted and treated as sss (the reference's HYB variant
  // asserts in its default multi-threaded build).
  static SparseMatrix<IndexT, ValueT> *create(const std::string &filename,
                                              Format format = Format::csr,
                                              Platform platform = Platform::gpu);
};

} // namespace sparse
} // namespace matrix
} // namespace cfs

#endif
