// Host-only sanitizer harness for the parallel Matrix-Market reader (src/mmf.cpp):
// loads every file given on the command line (golden fixtures and deliberately
// malformed inputs written by the test) under AddressSanitizer + UBSan.  A refusal
// with an error message is fine; a memory error is not.
#include <cstdio>
#include <cstdlib>

extern "C" {
int cfs_mmf_load_csr_f64(const char *path, int *nrows, int *ncols, long *nnz, int *symmetric,
                         int **rowptr, int **colind, double **values, char *err, int errlen);
void cfs_mmf_free(void *p);
}

int main(int argc, char **argv) {
  int loaded = 0, refused = 0;
  for (int k = 1; k < argc; k++) {
    int nr = 0, nc = 0, sym = 0, *rp = nullptr, *ci = nullptr;
    long nnz = 0;
    double *va = nullptr;
    char err[256] = {0};
    if (cfs_mmf_load_csr_f64(argv[k], &nr, &nc, &nnz, &sym, &rp, &ci, &va, err, sizeof err) == 0) {
      long touched = 0; // walk everything the loader returned
      for (int i = 0; i < nr; i++)
        for (int j = rp[i]; j < rp[i + 1]; j++) touched += ci[j] + (va[j] != 0.0);
      if (rp[nr] != nnz) return 3;
      (void)touched;
      cfs_mmf_free(rp);
      cfs_mmf_free(ci);
      cfs_mmf_free(va);
      loaded++;
    } else {
      refused++;
    }
  }
  printf("asan_reader: %d loaded, %d refused, OK\n", loaded, refused);
  return 0;
}
