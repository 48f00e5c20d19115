// matrix/sparse_matrix.hpp -- the abstract matrix callers hold, and its factory.
//
// Same class, member names and signatures as the reference
// (include/matrix/sparse_matrix.hpp:23-41, sparse_matrix.tpp:13-24), so that
// `SparseMatrix<int, double>::create(file, Format::sss)` compiles unchanged.
// What differs is where the work happens: the default Platform is gpu, tune()
// builds the MI355X tile schedule behind the C ABI (include/cfs_hip.h), and
// dense_vector_multiply() launches HIP kernels.
#ifndef CFS_SPARSE_MATRIX_HPP
#define CFS_SPARSE_MATRIX_HPP

#include <cstddef>
// <random> and <string> are part of what this header gives its includers in the
// reference (include/matrix/sparse_matrix.hpp:4-5): the unmodified drivers use
// random_device / mt19937 through it (bench_spmv_mmf.cpp:123-125).
#include <random>
#include <string>

#include "cfs_config.hpp"
#include "utils/platform.hpp"

namespace cfs {

using namespace util;

namespace matrix {
namespace sparse {

template <typename IndexT, typename ValueT> class SparseMatrix {
public:
  // ---- construction -----------------------------------------------------------
  // Matrix-Market file -> matrix (the only way the drivers build one).
  //   Format::csr   every stored entry, general CSR kernel
  //   Format::sss   lower triangle + diagonal when the FILE is symmetric; a general
  //                 file silently becomes csr (csr_matrix.tpp:13-19)
  //   Format::hyb   sss with the far entries kept apart (CFS_HIP_FLAG_HYB, src/csr.cpp;
  //                 the reference's split_by_bandwidth, csr_matrix.tpp:312-401, which
  //                 asserts in its default multi-threaded build -- a working feature here)
  //   Platform::cpu refused with an error: this build has no CPU path
  static SparseMatrix<IndexT, ValueT> *create(const std::string &filename,
                                              Format format = Format::csr,
                                              Platform platform = Platform::gpu);
  virtual ~SparseMatrix() {}

  // ---- shape and storage --------------------------------------------------------
  virtual int nrows() const = 0;
  virtual int ncols() const = 0;
  // stored entries with BOTH triangles expanded -- what the reference reports and
  // what the benchmark's flop count uses (bench_spmv_mmf.cpp:168)
  virtual int nnz() const = 0;
  // true only if the file was symmetric AND the symmetric format was asked for
  virtual bool symmetric() const = 0;
  // bytes held: host CSR before tune(), device memory of the schedule after it
  virtual size_t size() const = 0;
  virtual Platform platform() const = 0;

  // ---- the operator ---------------------------------------------------------------
  // builds the device schedule; SpDMV's constructor calls it.  Tuning::Aggressive
  // also runs the measured tuning steps (window shape, XCD shares), Tuning::None
  // builds the schedule once.
  virtual bool tune(Kernel k, Tuning t = Tuning::Aggressive) = 0;
  // y <- A x.  y is fully overwritten and needs no zeroing; x and y must not
  // alias.  Host pointers are staged through HBM on every call (slow, drop-in);
  // Platform::gpu pointers from internal_alloc stay resident (fast, enqueued).
  virtual void dense_vector_multiply(ValueT *__restrict y, const ValueT *__restrict x) = 0;
};

} // namespace sparse
} // namespace matrix
} // namespace cfs

#endif
