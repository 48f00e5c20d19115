// test_spmv_mmf -- self-check of the MI355X build, same protocol as the
// reference's only test (test/test_spmv_mmf.cpp:31-120): multiply twice with the
// requested format on an un-zeroed y, multiply once with plain CSR
// (Tuning::None), compare element-wise with isEqual, print PASSED!/FAILED!.
// Host pointers are used on purpose: this is the unmodified-caller path.
//     test_spmv_mmf <mmf_file> <format>(0: CSR, 1: CFS-SSS, 2: CFH-SSS)
#include <cstdlib>
#include <iostream>
#include <random>

#include "cfs.hpp"

using namespace std;
using namespace cfs::util;
using namespace cfs::util::memory;
using namespace cfs::matrix::sparse;
using namespace cfs::kernel::sparse;

typedef int INDEX;
typedef double VALUE;

int main(int argc, char **argv) {
  if (argc < 3) {
    cerr << "Error in number of arguments!" << endl;
    cout << "Usage: " << argv[0] << " <mmf_file> <format>(0: CSR, 1: CFS-SSS, 2: CFH-SSS)"
         << endl;
    return 1;
  }
  const string mmf_file(argv[1]);
  const int fmt = atoi(argv[2]);
  if (fmt < 0 || fmt > 2) {
    cerr << "Error in arguments!" << endl;
    return 1;
  }
  static const Format formats[] = {Format::csr, Format::sss, Format::hyb};
  SparseMatrix<INDEX, VALUE> *A = SparseMatrix<INDEX, VALUE>::create(mmf_file, formats[fmt]);
  const int M = A->nrows(), N = A->ncols();

  VALUE *x = (VALUE *)internal_alloc((size_t)N * sizeof(VALUE));
  VALUE *y = (VALUE *)internal_alloc((size_t)M * sizeof(VALUE));
  const char *seed_env = getenv("CFS_SEED");
  mt19937 gen(seed_env ? (unsigned)atoi(seed_env) : random_device()());
  uniform_real_distribution<> dis_val(10.01, 20.42);
  for (int i = 0; i < N; i++) x[i] = dis_val(gen);
  for (int i = 0; i < M; i++) y[i] = -12345.678; // never zeroed by the reference: poison it

  SpDMV<INDEX, VALUE> fn(A, Tuning::Aggressive);
  for (int i = 0; i < 2; ++i) fn(y, M, x, N);

  SparseMatrix<INDEX, VALUE> *A_test = SparseMatrix<INDEX, VALUE>::create(mmf_file, Format::csr);
  SpDMV<INDEX, VALUE> test(A_test, Tuning::None);
  VALUE *y_test = (VALUE *)internal_alloc((size_t)M * sizeof(VALUE));
  test(y_test, M, x, N);

  bool passed = true;
  for (INDEX i = 0; i < M; i++) {
    if (!isEqual(y[i], y_test[i])) {
      cout << "element " << i << " differs: " << y[i] << " vs " << y_test[i] << endl;
      passed = false;
      break;
    }
  }
  cout << (passed ? "PASSED!" : "FAILED!") << endl;

  delete A;
  delete A_test;
  internal_free(x);
  internal_free(y);
  internal_free(y_test);
  return passed ? 0 : 2;
}
