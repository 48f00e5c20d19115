// ref_mmf_harness.cpp -- OUR driver around the GENUINE reference MMF reader.
//
// TEST INFRASTRUCTURE ONLY (build container only; /root/reference does not
// exist on the GPU box and nothing here travels as source of the reference).
// oracle/Makefile compiles this file together with the reference's own
// src/mmf.cpp + include/io/mmf.hpp *where they lie* under /root/reference and
// puts the binary into oracle/_ref/ (git-ignored).  It is used to
//   (1) pin oracle/cfs_oracle.c's reader restatement (tests/test_oracle_ref.py),
//   (2) generate tests/golden/*.csr.npz (tests/golden/make_golden.py).
//
// usage: ref_mmf_dump <file.mtx> <f64|f32> <out.bin>
// out.bin: int32 nrows, ncols, nnz, symmetric; then nnz x {int32 row, int32 col}
//          (ONE-based, in the reader's sorted order) ; then nnz values.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "io/mmf.hpp"

template <typename V> static int dump(const char *path, const char *out) {
  cfs::io::MMF<int, V> mmf(path); // prints + exit(1) on malformed input
  int hdr[4] = {mmf.GetNrRows(), mmf.GetNrCols(), mmf.GetNrNonzeros(),
                mmf.IsSymmetric() ? 1 : 0};
  std::vector<int> rc;
  std::vector<V> val;
  rc.reserve(2 * (size_t)hdr[2]);
  val.reserve((size_t)hdr[2]);
  auto it = mmf.begin();
  auto end = mmf.end();
  for (; it != end; ++it) {
    rc.push_back((*it).row);
    rc.push_back((*it).col);
    val.push_back((*it).val);
  }
  FILE *f = fopen(out, "wb");
  if (!f) return 2;
  fwrite(hdr, sizeof(int), 4, f);
  fwrite(rc.data(), sizeof(int), rc.size(), f);
  fwrite(val.data(), sizeof(V), val.size(), f);
  fclose(f);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s <file.mtx> <f64|f32> <out.bin>\n", argv[0]);
    return 64;
  }
  if (!strcmp(argv[2], "f64")) return dump<double>(argv[1], argv[3]);
  return dump<float>(argv[1], argv[3]);
}
