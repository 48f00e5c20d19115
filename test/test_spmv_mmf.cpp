// test_spmv_mmf -- self-check of the MI355X build.
//
//     test_spmv_mmf <mmf_file> <format>(0: CSR, 1: CFS-SSS, 2: CFH-SSS)
//
// Protocol of the reference's only test (test/test_spmv_mmf.cpp:31-120): the
// matrix in the requested format multiplies a random vector twice into a y that
// was never zeroed; the same file as plain CSR (Tuning::None) multiplies it once;
// the two results must agree element-wise under isEqual.  Prints PASSED! /
// FAILED!.  Host vectors on purpose: this is what an unmodified caller does.
// CFS_SEED fixes the random vector (the reference draws from random_device).
#include <cstdlib>
#include <iostream>
#include <random>
#include <string>

#include "cfs.hpp"

namespace {

typedef int INDEX;
typedef double VALUE;
typedef cfs::matrix::sparse::SparseMatrix<INDEX, VALUE> Matrix;
typedef cfs::kernel::sparse::SpDMV<INDEX, VALUE> Multiply;

// a host vector from the library's allocator, released on scope exit
struct HostVector {
  VALUE *p;
  explicit HostVector(int n)
      : p((VALUE *)cfs::util::memory::internal_alloc((size_t)n * sizeof(VALUE))) {}
  ~HostVector() { cfs::util::memory::internal_free(p); }
  VALUE &operator[](int i) { return p[i]; }
};

bool parse_format(const char *arg, cfs::util::Format *out) {
  const int k = atoi(arg);
  if (k == 0) *out = cfs::util::Format::csr;
  else if (k == 1) *out = cfs::util::Format::sss;
  else if (k == 2) *out = cfs::util::Format::hyb;
  else return false;
  return true;
}

void fill_uniform(HostVector &v, int n, double lo, double hi) {
  const char *seed = getenv("CFS_SEED");
  std::mt19937 gen(seed ? (unsigned)atoi(seed) : std::random_device()());
  std::uniform_real_distribution<> draw(lo, hi);
  for (int i = 0; i < n; i++) v[i] = draw(gen);
}

// index of the first element the two vectors disagree on, or -1
int first_mismatch(HostVector &a, HostVector &b, int n) {
  for (int i = 0; i < n; i++)
    if (!cfs::util::isEqual(a[i], b[i])) return i;
  return -1;
}

} // namespace

int main(int argc, char **argv) {
  cfs::util::Format fmt;
  if (argc < 3) {
    std::cerr << "Error in number of arguments!" << std::endl;
    std::cout << "Usage: " << argv[0]
              << " <mmf_file> <format>(0: CSR, 1: CFS-SSS, 2: CFH-SSS)" << std::endl;
    return 1;
  }
  if (!parse_format(argv[2], &fmt)) {
    std::cerr << "Error in arguments!" << std::endl;
    return 1;
  }
  const std::string file(argv[1]);

  Matrix *A = Matrix::create(file, fmt);
  const int rows = A->nrows(), cols = A->ncols();
  HostVector x(cols), y(rows), y_csr(rows);
  fill_uniform(x, cols, 10.01, 20.42);
  for (int i = 0; i < rows; i++) y[i] = -12345.678; // the reference never zeroes y: poison it

  Multiply multiply(A, cfs::util::Tuning::Aggressive);
  multiply(y.p, rows, x.p, cols);
  multiply(y.p, rows, x.p, cols); // twice: the second call must not see the first

  Matrix *G = Matrix::create(file, cfs::util::Format::csr);
  Multiply ground_truth(G, cfs::util::Tuning::None);
  ground_truth(y_csr.p, rows, x.p, cols);

  const int bad = first_mismatch(y, y_csr, rows);
  if (bad >= 0)
    std::cout << "element " << bad << " differs: " << y[bad] << " vs " << y_csr[bad] << std::endl;
  std::cout << (bad < 0 ? "PASSED!" : "FAILED!") << std::endl;

  delete A;
  delete G;
  return bad < 0 ? 0 : 2;
}
