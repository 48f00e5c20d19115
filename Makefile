# Plain Makefile of the MI355X build (autotools are not installed in the build
# image; configure.ac / Makefile.am describe the same build for machines that
# have them).  Mirrors the reference's products: libsparse + bench_spmv_mmf +
# test_spmv_mmf.
#
#   make                   single precision drivers (the reference's default)
#   make DP=1              --enable-dp : -D_USE_DOUBLE
#   make LOG=1             --enable-log: -D_LOG_INFO
ROCM    ?= /opt/rocm
HIPCC   ?= $(ROCM)/bin/hipcc
CXX     ?= g++
ARCH    ?= gfx950
BUILD   ?= build
PKG     := cfs_spmv_amd
DEFS    := $(if $(DP),-D_USE_DOUBLE) $(if $(LOG),-D_LOG_INFO)
CXXFLAGS ?= -O2 -std=c++11 -fopenmp -Wall -fPIC
INC     := -Iinclude

all: $(PKG)/libcfs_hip.so $(BUILD)/libsparse.so $(BUILD)/bench_spmv_mmf $(BUILD)/test_spmv_mmf

# kernels + C ABI (also what cfs_spmv_amd/build.py builds)
$(PKG)/libcfs_hip.so: $(PKG)/csrc/cfs_hip.hip $(PKG)/csrc/cfs_plan.hpp $(PKG)/csrc/cfs_devplan.hpp $(PKG)/csrc/cfs_comm.hpp $(PKG)/csrc/cfs_csr.hpp $(PKG)/csrc/cfs_solver.hpp $(PKG)/csrc/cfs_runtime.hpp include/cfs_hip.h
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -fopenmp $(INC) -I$(PKG)/csrc $< -o $@ -ldl

LIBSRC := src/allocator.cpp src/runtime.cpp src/mmf.cpp src/csr.cpp src/cfs.cpp
$(BUILD)/libsparse.so: $(LIBSRC) $(wildcard include/*.hpp include/*/*.hpp) $(PKG)/libcfs_hip.so
	@mkdir -p $(BUILD)
	$(CXX) $(CXXFLAGS) $(DEFS) $(INC) -shared $(LIBSRC) -o $@ \
	    -L$(PKG) -lcfs_hip -Wl,-rpath,'$$ORIGIN/../$(PKG)'

$(BUILD)/bench_spmv_mmf: bench/bench_spmv_mmf.cpp $(BUILD)/libsparse.so
	$(CXX) $(CXXFLAGS) $(DEFS) $(INC) $< -o $@ -L$(BUILD) -lsparse -L$(PKG) -lcfs_hip \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,'$$ORIGIN/../$(PKG)'

# the self-check is double precision, like the reference's (typedef double VALUE)
$(BUILD)/test_spmv_mmf: test/test_spmv_mmf.cpp $(BUILD)/libsparse.so
	$(CXX) $(CXXFLAGS) $(INC) $< -o $@ -L$(BUILD) -lsparse -L$(PKG) -lcfs_hip \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,'$$ORIGIN/../$(PKG)'

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(BUILD)

.PHONY: all oracle clean
