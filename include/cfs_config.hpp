// cfs_config.hpp -- build configuration of the MI355X build of cfs-spmv.
//
// The reference pulls an autoheader-generated <config.h> here
// (include/cfs_config.hpp:4 there).  This build works with or without one:
// `./configure` (configure.ac) writes config.h; the plain Makefile passes the
// same macros (-D_USE_DOUBLE, -D_LOG_INFO) on the command line instead.
#ifndef CFS_CONFIG_HPP
#define CFS_CONFIG_HPP

#if defined(HAVE_CONFIG_H) || (defined(__has_include) && __has_include(<config.h>))
#if defined(__has_include)
#if __has_include(<config.h>)
#include <config.h>
#endif
#else
#include <config.h>
#endif
#endif

#endif
