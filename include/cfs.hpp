// cfs.hpp -- umbrella header of the MI355X build of cfs-spmv (same include set
// as the reference's include/cfs.hpp so that `#include "cfs.hpp"` keeps working).
#ifndef CFS_HPP
#define CFS_HPP

#include "cfs_config.hpp"
#include "utils/platform.hpp"
#include "utils/allocator.hpp"
#include "utils/runtime.hpp"
#include "matrix/sparse_matrix.hpp"
#include "matrix/csr_matrix.hpp"
#include "kernel/sparse_kernel.hpp"

#endif
