// matrix/csr_matrix.hpp -- CSR storage + the symmetric (SSS) device format behind
// it (reference: include/matrix/csr_matrix.hpp:47-75 for the public part).
//
// Everything the reference keeps private here (SymThreadData, conflict graph,
// colouring, the eleven CPU kernels) is replaced by an opaque handle of the C
// ABI (include/cfs_hip.h): tune() builds the MI355X tile schedule and uploads
// it, dense_vector_multiply() launches the HIP kernels.
#ifndef CFS_CSR_MATRIX_HPP
#define CFS_CSR_MATRIX_HPP

#include <cstddef>
#include <string>

#include "cfs_config.hpp"
#include "matrix/sparse_matrix.hpp"
#include "utils/allocator.hpp"
#include "utils/platform.hpp"
#include "utils/runtime.hpp"

namespace cfs {

using namespace util::memory;
using namespace util::runtime;

namespace matrix {
namespace sparse {

template <typename IndexT, typename ValueT>
class CSRMatrix : public SparseMatrix<IndexT, ValueT> {
public:
  CSRMatrix() = delete;
  // from a Matrix-Market file (owns its arrays)
  CSRMatrix(const std::string &filename, Platform platform = Platform::gpu,
            bool symmetric = false, bool hybrid = false);
  // around caller arrays: full CSR, 0-based, no ownership taken
  CSRMatrix(IndexT *rowptr, IndexT *colind, ValueT *values, IndexT nrows, IndexT ncols,
            bool symmetric = false, bool hybrid = false, Platform platform = Platform::gpu);
  virtual ~CSRMatrix();

  virtual int nrows() const override { return nrows_; }
  virtual int ncols() const override { return ncols_; }
  virtual int nnz() const override { return nnz_; }
  virtual bool symmetric() const override { return symmetric_; }
  virtual size_t size() const override;
  virtual Platform platform() const override { return platform_; }
  virtual bool tune(Kernel k, Tuning t) override;
  virtual void dense_vector_multiply(ValueT *__restrict y, const ValueT *__restrict x) override;

  // host CSR; for a symmetric matrix valid only before tune(), which releases
  // the full CSR exactly like the reference's compress_symmetry()
  IndexT *rowptr() const { return rowptr_; }
  IndexT *colind() const { return colind_; }
  ValueT *values() const { return values_; }

private:
  Platform platform_;
  int nrows_, ncols_, nnz_;
  bool symmetric_, owns_data_, hybrid_, tuned_;
  IndexT *rowptr_;
  IndexT *colind_;
  ValueT *values_;
  int nthreads_;       // CFS_NUM_THREADS latched at construction (reported only)
  void *sym_handle_;   // cfs_hip_sym_t
  void *csr_handle_;   // cfs_hip_csr_t
  size_t device_bytes_;
  void release_host_csr();
};

} // namespace sparse
} // namespace matrix
} // namespace cfs

#endif
