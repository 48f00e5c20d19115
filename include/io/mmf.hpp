// io/mmf.hpp -- Matrix-Market coordinate reader of the MI355X build.
//
// Produces the same CSR the reference's reader + CSRMatrix constructor produce
// (include/io/mmf.hpp:179-343, include/matrix/csr_matrix.tpp:74-107: symmetric
// files are expanded to both triangles, entries sorted by (row, col), explicit
// zeros kept, 2-token "pattern" lines get the value 0.42) but is built for
// files of hundreds of millions of entries: the file is mapped, cut at line
// boundaries and parsed by all host cores, then bucketed by row (counting
// sort) instead of comparison-sorted.  Values go through strtod, like atof.
//
// Deliberately more lenient than the reference reader: any run of blanks/tabs
// separates tokens, a last line without '\n' is read, '%' lines are skipped
// anywhere.  Every file the reference accepts yields bit-identical arrays.
#ifndef CFS_MMF_HPP
#define CFS_MMF_HPP

#include <string>
#include <vector>

namespace cfs {
namespace io {

template <typename IndexType, typename ValueType> struct CsrArrays {
  IndexType nrows = 0, ncols = 0;
  long nnz = 0; // expanded count
  bool symmetric = false;
  std::vector<IndexType> rowptr, colind;
  std::vector<ValueType> values;
};

// Returns false and fills `error` on malformed input (the reference prints and
// exits; the CSRMatrix constructor of this build does that with the message).
template <typename IndexType, typename ValueType>
bool LoadMmfCsr(const std::string &filename, CsrArrays<IndexType, ValueType> &out,
                std::string &error);

} // namespace io
} // namespace cfs

#endif
