# Builds the MI355X SpMV engine.  gfx950 only; hipcc cross-compiles without a GPU.
HIPCC      ?= /opt/rocm/bin/hipcc
CXX        ?= g++
ARCH       ?= gfx950
HIPFLAGS   ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Iinclude -Icask_amd/csrc -Wall -Wno-unused-function
LIBDIR     := cask_amd/lib

all: $(LIBDIR)/libcask_hip.so

$(LIBDIR)/libcask_hip.so: cask_amd/csrc/cask_hip.hip cask_amd/csrc/spmv_kernels.hpp cask_amd/csrc/blas1_kernels.hpp include/cask_hip.h
	mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ cask_amd/csrc/cask_hip.hip

oracle:
	$(MAKE) -C oracle _build/libcask_oracle.so

clean:
	rm -rf $(LIBDIR) build oracle/_build

.PHONY: all oracle clean
