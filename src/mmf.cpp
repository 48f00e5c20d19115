// mmf.cpp -- parallel Matrix-Market reader (see include/io/mmf.hpp for the
// contract and for what it mirrors in the reference).
#include "io/mmf.hpp"
#include "utils/runtime.hpp"

#include <fcntl.h>
#include <omp.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace cfs {
namespace io {

namespace {

struct Mapped {
  const char *p = nullptr;
  size_t n = 0;
  int fd = -1;
  ~Mapped() {
    if (p && n) munmap((void *)p, n);
    if (fd >= 0) close(fd);
  }
};

inline bool blank(char c) { return c == ' ' || c == '\t' || c == '\r'; }

// tokens of the line [b, e): up to 8 (begin, end) pairs
struct Tokens {
  const char *b[8], *e[8];
  int n = 0;
};
inline void tokenize(const char *b, const char *e, Tokens &t) {
  t.n = 0;
  while (b < e && t.n < 8) {
    while (b < e && blank(*b)) b++;
    if (b >= e) break;
    t.b[t.n] = b;
    while (b < e && !blank(*b)) b++;
    t.e[t.n++] = b;
  }
}
inline bool tok_is(const Tokens &t, int i, const char *s) {
  size_t l = strlen(s);
  return i < t.n && (size_t)(t.e[i] - t.b[i]) == l && memcmp(t.b[i], s, l) == 0;
}
// atoi on a token, bounded by the token's end: the file is mmap'ed, and a last
// token that runs to the end of a mapping whose size is a multiple of the page size
// has nothing behind it for strtol to stop at.  Like atoi: optional sign, digits,
// stops at the first other character.
inline long tok_long(const Tokens &t, int i) {
  const char *q = t.b[i], *e = t.e[i];
  bool neg = false;
  if (q < e && (*q == '-' || *q == '+')) neg = *q++ == '-';
  long v = 0;
  while (q < e && *q >= '0' && *q <= '9') v = v * 10 + (*q++ - '0');
  return neg ? -v : v;
}
// strtod on a copy of the token (NUL-terminated: see tok_long)
inline double tok_strtod(const char *b, const char *e) {
  char buf[128];
  const size_t l = (size_t)(e - b);
  if (l < sizeof buf) {
    memcpy(buf, b, l);
    buf[l] = 0;
    return strtod(buf, nullptr);
  }
  std::string tmp(b, e);
  return strtod(tmp.c_str(), nullptr);
}
// atof on a token, bit-exact with strtod: Clinger's fast path -- at most 15
// significant digits (mantissa < 2^53) and a decimal exponent within +-22 are ONE
// correctly rounded multiplication / division of two exact doubles -- covers what
// SuiteSparse files usually hold; everything else (17-digit values, huge or tiny
// exponents, inf / nan / hex) goes to strtod itself.
inline double tok_double(const Tokens &t, int i) {
  static const double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,
                                 1e8,  1e9,  1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  const char *b = t.b[i], *e = t.e[i], *q = b;
  bool neg = false;
  if (q < e && (*q == '-' || *q == '+')) neg = *q++ == '-';
  unsigned long long m = 0;
  int digits = 0, exp10 = 0;
  bool any = false, ok = true;
  while (q < e && *q >= '0' && *q <= '9') {
    if (m || *q != '0') digits++;
    if (digits <= 15) m = m * 10 + (unsigned)(*q - '0');
    else ok = false;
    any = true;
    q++;
  }
  if (q < e && *q == '.') {
    q++;
    while (q < e && *q >= '0' && *q <= '9') {
      if (m || *q != '0') digits++;
      if (digits <= 15) {
        m = m * 10 + (unsigned)(*q - '0');
        exp10--;
      } else {
        ok = false;
      }
      any = true;
      q++;
    }
  }
  if (any && q < e && (*q == 'e' || *q == 'E')) {
    const char *r = q + 1;
    bool eneg = false;
    if (r < e && (*r == '-' || *r == '+')) eneg = *r++ == '-';
    int ev = 0, nd = 0;
    while (r < e && *r >= '0' && *r <= '9' && nd < 5) {
      ev = ev * 10 + (*r - '0');
      r++;
      nd++;
    }
    if (nd == 0 || (r < e && *r >= '0' && *r <= '9')) ok = false;
    exp10 += eneg ? -ev : ev;
    q = r;
  }
  if (!any || !ok || q != e || exp10 < -22 || exp10 > 22) return tok_strtod(b, e);
  double v = (double)m; // exact: m < 10^15 < 2^53
  v = exp10 < 0 ? v / p10[-exp10] : v * p10[exp10];
  return neg ? -v : v;
}

struct Elem {
  int row, col; // zero-based here
  double val;
  long seq;
};

} // namespace

// ---- optional binary cache ------------------------------------------------------
// Parsing Queen_4147 (~5 GB of text) dominates every run of the driver, so the
// CSR can be kept as a side file: set CFS_MTX_CACHE_DIR to a writable directory
// and <dir>/<basename>.<f32|f64>.csrbin is written after the first parse and
// mapped on later runs.  The cache is keyed by the source file's size and
// modification time; anything that does not match is ignored and rewritten.
struct CacheHeader {
  char magic[8]; // "CFSCSR1\0"
  int64_t src_size, src_mtime_ns, nrows, ncols, nnz;
  int32_t symmetric, value_bytes;
};

static std::string cache_path(const std::string &filename, size_t value_bytes) {
  const char *dir = getenv("CFS_MTX_CACHE_DIR");
  if (!dir || !*dir) return std::string();
  size_t slash = filename.find_last_of('/');
  std::string base = slash == std::string::npos ? filename : filename.substr(slash + 1);
  return std::string(dir) + "/" + base + (value_bytes == 8 ? ".f64" : ".f32") + ".csrbin";
}

template <typename IndexType, typename ValueType>
static bool cache_load(const std::string &path, const struct stat &src,
                       CsrArrays<IndexType, ValueType> &out) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  CacheHeader h;
  bool ok = fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, "CFSCSR1", 8) == 0 &&
            h.src_size == (int64_t)src.st_size &&
            h.src_mtime_ns == (int64_t)src.st_mtim.tv_sec * 1000000000LL + src.st_mtim.tv_nsec &&
            h.value_bytes == (int32_t)sizeof(ValueType) && h.nrows >= 0 && h.nnz >= 0;
  if (ok) {
    out.nrows = (IndexType)h.nrows;
    out.ncols = (IndexType)h.ncols;
    out.nnz = h.nnz;
    out.symmetric = h.symmetric != 0;
    out.rowptr.resize((size_t)h.nrows + 1);
    out.colind.resize((size_t)h.nnz);
    out.values.resize((size_t)h.nnz);
    ok = fread(out.rowptr.data(), sizeof(IndexType), out.rowptr.size(), f) == out.rowptr.size() &&
         fread(out.colind.data(), sizeof(IndexType), out.colind.size(), f) == out.colind.size() &&
         fread(out.values.data(), sizeof(ValueType), out.values.size(), f) == out.values.size();
  }
  fclose(f);
  return ok;
}

template <typename IndexType, typename ValueType>
static void cache_store(const std::string &path, const struct stat &src,
                        const CsrArrays<IndexType, ValueType> &a) {
  const std::string tmp = path + ".tmp";
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return; // the cache is best effort
  CacheHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, "CFSCSR1", 8);
  h.src_size = (int64_t)src.st_size;
  h.src_mtime_ns = (int64_t)src.st_mtim.tv_sec * 1000000000LL + src.st_mtim.tv_nsec;
  h.nrows = a.nrows;
  h.ncols = a.ncols;
  h.nnz = a.nnz;
  h.symmetric = a.symmetric ? 1 : 0;
  h.value_bytes = (int32_t)sizeof(ValueType);
  bool ok = fwrite(&h, sizeof h, 1, f) == 1 &&
            fwrite(a.rowptr.data(), sizeof(IndexType), a.rowptr.size(), f) == a.rowptr.size() &&
            fwrite(a.colind.data(), sizeof(IndexType), a.colind.size(), f) == a.colind.size() &&
            fwrite(a.values.data(), sizeof(ValueType), a.values.size(), f) == a.values.size();
  ok = fclose(f) == 0 && ok;
  if (ok) ok = rename(tmp.c_str(), path.c_str()) == 0;
  if (!ok) remove(tmp.c_str());
}

template <typename IndexType, typename ValueType>
static bool ParseMmfCsr(const std::string &filename, CsrArrays<IndexType, ValueType> &out,
                        std::string &error);

template <typename IndexType, typename ValueType>
bool LoadMmfCsr(const std::string &filename, CsrArrays<IndexType, ValueType> &out,
                std::string &error) {
  const std::string cpath = cache_path(filename, sizeof(ValueType));
  struct stat st;
  const bool have_stat = stat(filename.c_str(), &st) == 0;
  if (!cpath.empty() && have_stat && cache_load<IndexType, ValueType>(cpath, st, out)) return true;
  if (!ParseMmfCsr<IndexType, ValueType>(filename, out, error)) return false;
  if (!cpath.empty() && have_stat) cache_store<IndexType, ValueType>(cpath, st, out);
  return true;
}

template <typename IndexType, typename ValueType>
static bool ParseMmfCsr(const std::string &filename, CsrArrays<IndexType, ValueType> &out,
                        std::string &error) {
  Mapped m;
  m.fd = open(filename.c_str(), O_RDONLY);
  if (m.fd < 0) {
    error = "MMF file error.";
    return false;
  }
  struct stat st;
  if (fstat(m.fd, &st) != 0 || st.st_size == 0) {
    error = "MMF file error.";
    return false;
  }
  m.n = (size_t)st.st_size;
  m.p = (const char *)mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, m.fd, 0);
  if (m.p == MAP_FAILED) {
    m.p = nullptr;
    error = "MMF file error.";
    return false;
  }
  const char *p = m.p, *end = m.p + m.n;
  auto next_line = [&](const char *&b, const char *&e) -> bool {
    if (p >= end) return false;
    b = p;
    const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
    e = nl ? nl : end;
    p = nl ? nl + 1 : end;
    return true;
  };

  // ---- banner -----------------------------------------------------------------
  const char *lb, *le;
  Tokens t;
  bool symmetric = false, zero_based = false;
  if (!next_line(lb, le)) {
    error = "MMF file error.";
    return false;
  }
  tokenize(lb, le, t);
  bool size_line_pending = false;
  if (t.n > 0 && tok_is(t, 0, "%%MatrixMarket")) {
    if (t.n < 5) {
      error = "less arguments in header line of MMF file.";
      return false;
    }
    if (!tok_is(t, 2, "coordinate")) {
      error = "unsupported matrix format in header line of MMF file.";
      return false;
    }
    if (tok_is(t, 4, "general")) symmetric = false;
    else if (tok_is(t, 4, "symmetric")) symmetric = true;
    else {
      error = "unsupported symmetry in header line of MMF file.";
      return false;
    }
    for (int i = 5; i < t.n; i++) {
      if (tok_is(t, i, "base-0")) zero_based = true;
      else if (tok_is(t, i, "base-1")) zero_based = false;
    }
  } else if (t.n > 0 && (t.e[0] - t.b[0]) > 2 && t.b[0][0] == '%' && t.b[0][1] == '%') {
    error = "invalid header line in MMF file.";
    return false;
  } else if (t.n > 0 && t.b[0][0] != '%') {
    size_line_pending = true; // no banner: this IS the size line
  }
  // ---- size line ----------------------------------------------------------------
  if (!size_line_pending) {
    for (;;) {
      if (!next_line(lb, le)) {
        error = "size line error in MMF file.";
        return false;
      }
      tokenize(lb, le, t);
      if (t.n > 0 && t.b[0][0] != '%') break;
    }
  }
  if (t.n < 2) {
    error = "bad input, less arguments in line of MMF file.";
    return false;
  }
  const long nrows = tok_long(t, 0), ncols = tok_long(t, 1);
  const long declared = t.n >= 3 ? (long)tok_double(t, 2) : 0;
  if (nrows < 0 || ncols < 0 || declared < 0 || nrows > 0x7fffffffL || ncols > 0x7fffffffL) {
    error = "bad size line in MMF file.";
    return false;
  }

  // CFS_MMF_VERBOSE=1 prints where the reader spends its time
  const bool verbose = getenv("CFS_MMF_VERBOSE") != nullptr;
  double t_last = omp_get_wtime();
  auto lap = [&](const char *what) {
    if (!verbose) return;
    const double now = omp_get_wtime();
    fprintf(stderr, "[cfs_mmf] %-28s %8.3f s\n", what, now - t_last);
    t_last = now;
  };
  // ---- entries: cut the body at line boundaries, parse in parallel --------------
  const char *body = p;
  const int nth = std::max(1, cfs::util::runtime::get_host_threads());
  std::vector<const char *> cut(nth + 1, end);
  cut[0] = body;
  for (int i = 1; i < nth; i++) {
    const char *q = body + (size_t)(end - body) * i / nth;
    if (q < cut[i - 1]) q = cut[i - 1];
    const char *nl = (const char *)memchr(q, '\n', (size_t)(end - q));
    cut[i] = nl ? nl + 1 : end;
  }
  std::vector<std::vector<Elem>> part(nth);
  std::vector<int> bad(nth, 0);
#pragma omp parallel num_threads(nth)
  {
    const int tid = omp_get_thread_num();
    std::vector<Elem> &v = part[tid];
    const char *q = cut[tid], *qe = cut[tid + 1];
    v.reserve((size_t)(qe - q) / 24 + 16);
    Tokens tk;
    while (q < qe) {
      const char *nl = (const char *)memchr(q, '\n', (size_t)(qe - q));
      const char *e = nl ? nl : qe;
      tokenize(q, e, tk);
      q = nl ? nl + 1 : qe;
      if (tk.n == 0 || tk.b[0][0] == '%') continue;
      if (tk.n < 2) {
        bad[tid] = 1;
        break;
      }
      Elem el;
      el.row = (int)tok_long(tk, 0) - (zero_based ? 0 : 1);
      el.col = (int)tok_long(tk, 1) - (zero_based ? 0 : 1);
      el.val = tk.n >= 3 ? tok_double(tk, 2) : 0.42; // pattern entries
      el.seq = 0;
      v.push_back(el);
    }
  }
  lap("parse (parallel)");
  long read = 0;
  for (int i = 0; i < nth; i++) {
    if (bad[i]) {
      error = "bad input, less arguments in line of MMF file.";
      return false;
    }
    read += (long)part[i].size();
  }
  if (read < declared) {
    error = "Requesting dereference, but mmf ended.";
    return false;
  }
  // the reference reads exactly `declared` entries and ignores the rest
  long keep = declared;
  // ---- expand, bucket by row (counting sort), order each row by column ----------
  // Entries beyond the declared count are ignored: trim the parts first, so that
  // both passes below can run one thread per part.  `base[i]` = entries of the
  // parts before i (the file order of an entry = base + its index in its part).
  std::vector<long> base(nth + 1, 0);
  for (int i = 0; i < nth; i++) {
    const long room = keep - base[i];
    if ((long)part[i].size() > room) part[i].resize((size_t)std::max(0L, room));
    base[i + 1] = base[i] + (long)part[i].size();
  }
  std::vector<long> rowcnt((size_t)nrows + 1, 0);
  bool range_error = false;
#pragma omp parallel for schedule(static, 1) num_threads(nth)
  for (int i = 0; i < nth; i++)
    for (size_t k = 0; k < part[i].size(); k++) {
      const Elem &el = part[i][k];
      if (el.row < 0 || el.row >= nrows || el.col < 0 || el.col >= ncols ||
          (symmetric && el.row != el.col && (el.col >= nrows || el.row >= ncols))) {
#pragma omp atomic write
        range_error = true;
        continue;
      }
#pragma omp atomic
      rowcnt[el.row + 1]++;
      if (symmetric && el.row != el.col) {
#pragma omp atomic
        rowcnt[el.col + 1]++;
      }
    }
  if (range_error) {
    error = "entry out of range in MMF file.";
    return false;
  }
  for (long r = 0; r < nrows; r++) rowcnt[r + 1] += rowcnt[r];
  lap("count rows");
  const long nnz = rowcnt[nrows];
  if (nnz > 0x7fffffffL) {
    error = "more than 2^31-1 nonzeros: int indices are API (src/csr.cpp)";
    return false;
  }
  struct CV {
    int col;
    double val;
    long seq;
  };
  std::vector<CV> buf((size_t)nnz);
  {
    // a row's bucket is filled in whatever order the threads arrive; `seq` (twice
    // the file order, +1 for the mirrored copy) restores the file order of equal
    // columns in the sort below, so the result does not depend on the threads
    std::vector<long> fill(rowcnt.begin(), rowcnt.end() - 1);
#pragma omp parallel for schedule(static, 1) num_threads(nth)
    for (int i = 0; i < nth; i++)
      for (size_t k = 0; k < part[i].size(); k++) {
        const Elem &el = part[i][k];
        const long seq = 2 * (base[i] + (long)k);
        long pos;
#pragma omp atomic capture
        pos = fill[el.row]++;
        buf[pos] = CV{el.col, el.val, seq};
        if (symmetric && el.row != el.col) {
#pragma omp atomic capture
          pos = fill[el.col]++;
          buf[pos] = CV{el.row, el.val, seq + 1};
        }
      }
    for (auto &v : part) std::vector<Elem>().swap(v);
  }
  lap("bucket by row");
  out.nrows = (IndexType)nrows;
  out.ncols = (IndexType)ncols;
  out.nnz = nnz;
  out.symmetric = symmetric;
  out.rowptr.resize((size_t)nrows + 1);
  out.colind.resize((size_t)nnz);
  out.values.resize((size_t)nnz);
#pragma omp parallel for schedule(dynamic, 1024) num_threads(nth)
  for (long r = 0; r < nrows; r++) {
    CV *b = buf.data() + rowcnt[r], *e = buf.data() + rowcnt[r + 1];
    // (col, input order): duplicates keep their file order
    std::sort(b, e, [](const CV &x, const CV &y) {
      return x.col != y.col ? x.col < y.col : x.seq < y.seq;
    });
    for (CV *q = b; q < e; q++) {
      out.colind[q - buf.data()] = (IndexType)q->col;
      out.values[q - buf.data()] = (ValueType)q->val;
    }
  }
  for (long r = 0; r <= nrows; r++) out.rowptr[r] = (IndexType)rowcnt[r];
  lap("sort rows + emit");
  // the reference asserts that the last row is non-empty (csr_matrix.tpp:104)
  if (nrows > 0 && rowcnt[nrows] == rowcnt[nrows - 1] && nnz > 0) {
    // accepted here: trailing empty rows simply repeat rowptr
  }
  return true;
}

template bool LoadMmfCsr<int, float>(const std::string &, CsrArrays<int, float> &,
                                     std::string &);
template bool LoadMmfCsr<int, double>(const std::string &, CsrArrays<int, double> &,
                                      std::string &);

} // namespace io
} // namespace cfs

// C entry points for tests and foreign callers: the loader alone, no device.
// Arrays are malloc'ed; release them with cfs_mmf_free.
extern "C" {
int cfs_mmf_load_csr_f64(const char *path, int *nrows, int *ncols, long *nnz, int *symmetric,
                         int **rowptr, int **colind, double **values, char *err, int errlen) {
  cfs::io::CsrArrays<int, double> a;
  std::string e;
  if (!cfs::io::LoadMmfCsr<int, double>(path, a, e)) {
    if (err && errlen > 0) {
      strncpy(err, e.c_str(), (size_t)errlen - 1);
      err[errlen - 1] = 0;
    }
    return -1;
  }
  *nrows = a.nrows;
  *ncols = a.ncols;
  *nnz = a.nnz;
  *symmetric = a.symmetric ? 1 : 0;
  *rowptr = (int *)malloc(sizeof(int) * a.rowptr.size());
  *colind = (int *)malloc(sizeof(int) * (a.colind.size() + 1));
  *values = (double *)malloc(sizeof(double) * (a.values.size() + 1));
  if (!a.rowptr.empty()) memcpy(*rowptr, a.rowptr.data(), sizeof(int) * a.rowptr.size());
  if (!a.colind.empty()) memcpy(*colind, a.colind.data(), sizeof(int) * a.colind.size());
  if (!a.values.empty()) memcpy(*values, a.values.data(), sizeof(double) * a.values.size());
  return 0;
}
void cfs_mmf_free(void *p) { free(p); }
}
