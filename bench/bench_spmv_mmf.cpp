// bench_spmv_mmf -- benchmark driver of the MI355X build.
//
// Same command line and same result line as the reference driver
// (bench/bench_spmv_mmf.cpp:41-45, :139-173):
//     bench_spmv_mmf <mmf_file> <format>(0: CSR, 1: SSS, 2: HYB) <iterations>
//     matrix: <file> format: <CSR|SSS|HYB> preproc(sec): .. t(sec): .. gflops/s: ..
//     threads: .. size(MB): ..            [+ gbytes/s: .. hbm_pct: .. gpus: .. devices: ..]
// gflops/s = loops * 2 * nnz / t with nnz the expanded count (:168).
// x and y live in Platform::gpu memory during the timed loop (SURVEY.md
// section 7, "x/y residency"): they are filled on the host exactly as the
// reference does and moved once with internal_copy().  Formats 3 (MKL) and 4
// (RSB) of the reference are comparators for absent libraries and are refused.
#include <libgen.h>
#include <omp.h>

#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <random>

#include "cfs.hpp"

using namespace std;
using namespace cfs::util;
using namespace cfs::util::memory;
using namespace cfs::util::runtime;
using namespace cfs::matrix::sparse;
using namespace cfs::kernel::sparse;

typedef int INDEX;
#ifdef _USE_DOUBLE
typedef double VALUE;
#else
typedef float VALUE; // the reference's default build is single precision too
#endif

static void usage(const char *prog) {
  cout << "Usage: " << prog
       << " <mmf_file> <format>(0: CSR, 1:SSS, 2: HYB) <iterations>" << endl;
}

int main(int argc, char **argv) {
  if (argc < 4) {
    cerr << "Error in number of arguments!" << endl;
    usage(argv[0]);
    return 1;
  }
  const string mmf_file(argv[1]);
  const int fmt = atoi(argv[2]);
  if (fmt < 0 || fmt > 2) {
    cerr << "Error in arguments!" << endl;
    usage(argv[0]);
    return 1;
  }
  const size_t loops = (size_t)atoi(argv[3]);
  const int nthreads = (int)get_num_threads();
  // CFS_NUM_GPUS shards a symmetric (SSS / HYB) matrix; the general CSR kernel runs on one
  const int ngpus = fmt == 0 ? 1 : get_num_gpus();
  static const Format formats[] = {Format::csr, Format::sss, Format::hyb};
  static const char *names[] = {"CSR", "SSS", "HYB"};

  SparseMatrix<INDEX, VALUE> *A = SparseMatrix<INDEX, VALUE>::create(mmf_file, formats[fmt]);
  const int M = A->nrows(), N = A->ncols(), nnz = A->nnz();

  // x in U(0.01, 0.42) like the reference (:125); seedable for reproducible runs
  const char *seed_env = getenv("CFS_SEED");
  mt19937 gen(seed_env ? (unsigned)atoi(seed_env) : random_device()());
  uniform_real_distribution<> dis_val(0.01, 0.42);
  VALUE *x_host = (VALUE *)internal_alloc((size_t)N * sizeof(VALUE), Platform::cpu);
  VALUE *y_host = (VALUE *)internal_alloc((size_t)M * sizeof(VALUE), Platform::cpu);
  for (int i = 0; i < N; i++) x_host[i] = (VALUE)dis_val(gen);
  for (int i = 0; i < M; i++) y_host[i] = 0.0;
  VALUE *x = (VALUE *)internal_alloc((size_t)N * sizeof(VALUE), Platform::gpu);
  VALUE *y = (VALUE *)internal_alloc((size_t)M * sizeof(VALUE), Platform::gpu);
  internal_copy(x, Platform::gpu, x_host, Platform::cpu, (size_t)N * sizeof(VALUE));
  internal_copy(y, Platform::gpu, y_host, Platform::cpu, (size_t)M * sizeof(VALUE));

  double tstart = omp_get_wtime();
  SpDMV<INDEX, VALUE> spdmv(A); // tune(): host schedule + upload
  const double preproc_time = omp_get_wtime() - tstart;

#ifdef _LOG_INFO
  cout << "[INFO]: warming up caches..." << endl;
#endif
  for (size_t i = 0; i < loops / 2; i++) spdmv(y, M, x, N);
  synchronize();
#ifdef _LOG_INFO
  cout << "[INFO]: benchmarking SpDMV using " << names[fmt] << "..." << endl;
#endif
  tstart = omp_get_wtime();
  for (size_t i = 0; i < loops; i++) spdmv(y, M, x, N); // enqueued: x, y are device-resident
  synchronize();                                        // ... so wait before stopping the clock
  const double compute_time = omp_get_wtime() - tstart;
  internal_copy(y_host, Platform::cpu, y, Platform::gpu, (size_t)M * sizeof(VALUE));

  const double gflops = ((double)loops * 2 * nnz * 1.e-9) / compute_time;
  // algorithmic bytes, SURVEY.md 8(d): symmetric nnz_low*(4+s) + n*(4+3s)
  const double s = sizeof(VALUE);
  const double nnz_low = A->symmetric() ? (nnz - M) / 2.0 : (double)nnz;
  const double bytes = A->symmetric() ? nnz_low * (4 + s) + M * (4 + 3 * s)
                                      : nnz * (4 + s) + M * (4 + s) + N * s;
  const double gbs = bytes * loops * 1.e-9 / compute_time;
  // the peak is that of the DISTINCT devices the shards run on (several shards may share a
  // device: CFS_NUM_GPUS=4 on a one-GPU box is four shards of one 8 TB/s memory)
  const int visible = get_num_devices();
  const int ndev = ngpus < 1 ? 1 : (visible > 0 && ngpus > visible ? visible : ngpus);
  char *path = strdup(mmf_file.c_str());
  cout << setprecision(4) << "matrix: " << basename(path) << " format: " << names[fmt]
       << " preproc(sec): " << preproc_time << " t(sec): " << compute_time / loops
       << " gflops/s: " << gflops << " threads: " << nthreads
       << " size(MB): " << A->size() / (float)(1024 * 1024) << " gbytes/s: " << gbs
       << " hbm_pct: " << 100.0 * gbs / (8000.0 * ndev) << " gpus: " << ngpus << " devices: " << ndev << endl;
  free(path);

  delete A;
  internal_free(x, Platform::gpu);
  internal_free(y, Platform::gpu);
  internal_free(x_host, Platform::cpu);
  internal_free(y_host, Platform::cpu);
  return 0;
}
