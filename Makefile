# Builds the MI355X SpMV engine.  gfx950 only; hipcc cross-compiles without a GPU.
HIPCC      ?= /opt/rocm/bin/hipcc
CXX        ?= g++
ARCH       ?= gfx950
HIPFLAGS   ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -pthread -Iinclude -Icask_amd/csrc -Wall -Wno-unused-function
LIBDIR     := cask_amd/lib

GENDIR     := $(LIBDIR)/lib-generated
CXXFLAGS   ?= -std=c++14 -O2 -fPIC -Wall -Iinclude
HOSTSRC    := cask_amd/csrc/host/CaskHost.cpp
HOSTHDR    := $(wildcard include/cask/*.hpp) include/cask_hip.h
RPATHS     := -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,'$$ORIGIN/lib-generated' -Wl,-rpath,'$$ORIGIN/../cask_amd/lib' -Wl,-rpath,'$$ORIGIN/../cask_amd/lib/lib-generated'

all: $(LIBDIR)/libcask_hip.so $(LIBDIR)/libCaskHip.so $(GENDIR)/libSpmv_hip.so build/main build/test_host

# C++ host runtime over the C ABI (the SparkCpuLib of this build)
$(LIBDIR)/libCaskHip.so: $(HOSTSRC) $(HOSTHDR) $(LIBDIR)/libcask_hip.so
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOSTSRC) -L$(LIBDIR) -lcask_hip $(RPATHS)

# generated implementation library (target hip): loader constructor + design points
$(GENDIR)/libSpmv_hip.so: tools/gen_impl.py include/cask/GeneratedImplSupport.hpp
	python3 tools/gen_impl.py --out-dir $(GENDIR) --cpp $(CXX)

# DSE executable: build/main <bench-path> <params.json> -> dse_out.json
build/main: cask_amd/csrc/host/dse_main.cpp $(LIBDIR)/libCaskHip.so
	mkdir -p build
	$(CXX) $(CXXFLAGS) -o $@ $< -L$(LIBDIR) -lCaskHip -lcask_hip $(RPATHS)

# CPU unit tests of the host surface
build/test_host: tests/cpp/test_host.cpp $(HOSTHDR)
	mkdir -p build
	$(CXX) $(CXXFLAGS) -o $@ $<

# integration client (links the CPU oracle: test infrastructure only)
build/test_spmv_hip: tests/clients/test_spmv_client.cpp $(LIBDIR)/libCaskHip.so $(GENDIR)/libSpmv_hip.so oracle
	mkdir -p build
	$(CXX) $(CXXFLAGS) -o $@ $< -L$(LIBDIR) -L$(GENDIR) -Loracle/_build -lCaskHip -lSpmv_hip -lcask_hip -lcask_oracle \
	  $(RPATHS) -Wl,-rpath,'$$ORIGIN/../oracle/_build'

build/test_precond_hip: tests/clients/test_precond_client.cpp $(LIBDIR)/libCaskHip.so
	mkdir -p build
	$(CXX) $(CXXFLAGS) -o $@ $< -L$(LIBDIR) -lCaskHip -lcask_hip $(RPATHS)

build/test_bicg_hip: tests/clients/test_bicg_client.cpp $(LIBDIR)/libCaskHip.so
	mkdir -p build
	$(CXX) $(CXXFLAGS) -o $@ $< -L$(LIBDIR) -lCaskHip -lcask_hip $(RPATHS)

build/test_context_hip: tests/clients/test_context_client.cpp $(LIBDIR)/libCaskHip.so $(GENDIR)/libSpmv_hip.so
	mkdir -p build
	$(CXX) $(CXXFLAGS) -o $@ $< -L$(LIBDIR) -L$(GENDIR) -lCaskHip -lSpmv_hip -lcask_hip $(RPATHS)

# plain C over the C ABI alone: a row-sharded solve with the engine's RCCL collectives
build/test_sharded_solver_hip: tests/clients/test_sharded_solver_client.c $(LIBDIR)/libcask_hip.so include/cask_hip.h include/cask_hip_rccl.h
	mkdir -p build
	gcc -std=c11 -O2 -Wall -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include -o $@ $< -L$(LIBDIR) -lcask_hip -L/opt/rocm/lib -lamdhip64 -lm \
	  $(RPATHS) -Wl,-rpath,/opt/rocm/lib

clients: build/test_spmv_hip build/test_precond_hip build/test_bicg_hip build/test_context_hip build/test_sharded_solver_hip

# libcask_hip.so: one object per translation unit so that `make -j` compiles the merge-kernel
# instantiations (merge_ipt<N>.hip, the slow part) in parallel
ENGINESRC  := cask_hip cask_hip_dfe cask_hip_p2p cask_hip_precond cask_hip_rccl scan_launch merge_ipt2 merge_ipt4 merge_ipt8 merge_ipt16
ENGINEOBJ  := $(ENGINESRC:%=build/obj/%.o)
ENGINEHDR  := $(wildcard cask_amd/csrc/*.hpp) include/cask_hip.h include/cask_hip_dfe.h include/cask_hip_p2p.h include/cask_hip_rccl.h
# header dependencies come from the compiler (-MMD): touching scan_kernel.hpp does not recompile the merge kernels
build/obj/%.o: cask_amd/csrc/%.hip
	mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -MMD -MP -c -o $@ $<
-include $(ENGINEOBJ:.o=.d)
$(LIBDIR)/libcask_hip.so: $(ENGINEOBJ)
	mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -Wl,-s -o $@ $(ENGINEOBJ)   # (-s: no static symbol table -- 0.2 MB of template names; the C ABI is in .dynsym)

oracle:
	$(MAKE) -C oracle _build/libcask_oracle.so

# development microbenchmarks (tools/profile_round.sh uses membench as the FETCH_SIZE calibration kernel)
build/membench: tools/membench.hip
	mkdir -p build
	$(HIPCC) -O3 --offload-arch=$(ARCH) -o $@ $<
build/membench2: tools/membench2.hip
	mkdir -p build
	$(HIPCC) -O3 --offload-arch=$(ARCH) -o $@ $<
# round-3 microbenchmarks: one-launch streaming floor, the SCAN block shape, grid-wide hand-off, gather flavours
build/streamfloor build/scanfloor build/gatherbench build/lds_granule: build/%: tools/%.hip
	mkdir -p build
	$(HIPCC) -O3 --offload-arch=$(ARCH) -o $@ $<
build/gridsync: tools/gridsync_bench.hip
	mkdir -p build
	$(HIPCC) -O3 --offload-arch=$(ARCH) -o $@ $<
microbench: build/membench build/membench2 build/streamfloor build/scanfloor build/gatherbench build/gridsync

clean:
	rm -rf $(LIBDIR) build oracle/_build

.PHONY: all oracle clean clients microbench

# diagnostic build with in-kernel phase stamps (tools/stamps.py); never used by tests or bench
build/libcask_hip_stamps.so: cask_amd/csrc/cask_hip.hip $(ENGINEHDR)
	mkdir -p build
	$(HIPCC) $(HIPFLAGS) -DCASK_STAMPS -DCASK_UNITY -shared -o $@ cask_amd/csrc/cask_hip.hip cask_amd/csrc/cask_hip_dfe.hip cask_amd/csrc/cask_hip_p2p.hip cask_amd/csrc/cask_hip_precond.hip cask_amd/csrc/cask_hip_rccl.hip

# diagnostic builds of the merge kernel (development only; cask_amd/csrc/diag.hpp has the switches): build/libcask_hip_diag<N>.so
build/libcask_hip_diag%.so: cask_amd/csrc/cask_hip.hip $(ENGINEHDR)
	mkdir -p build
	$(HIPCC) $(HIPFLAGS) -DCASK_DIAG=$* -DCASK_UNITY -shared -o $@ cask_amd/csrc/cask_hip.hip cask_amd/csrc/cask_hip_dfe.hip cask_amd/csrc/cask_hip_p2p.hip cask_amd/csrc/cask_hip_precond.hip cask_amd/csrc/cask_hip_rccl.hip

# row f3: MatrixMarket ingest timing (host only)
build/ingest_time: tools/ingest_time.cpp $(HOSTHDR)
	mkdir -p build
	$(CXX) $(CXXFLAGS) -O2 -o $@ $<

# CPU sanitizers on everything that is host code: the host surface's unit tests, the ingest timer,
# the oracle's C restatement and the launch planners (plan_host.hpp, trsv_lanes_plan.hpp: host-only headers) under
# AddressSanitizer + UndefinedBehaviorSanitizer.  CPU build only -- no GPU-side sanitizer, no xnack+ code objects.
SANFLAGS := -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=all -Wall -Iinclude -Icask_amd/csrc
build/asan/test_host: tests/cpp/test_host.cpp $(HOSTHDR)
	mkdir -p build/asan
	$(CXX) $(SANFLAGS) -o $@ $<
build/asan/ingest_time: tools/ingest_time.cpp $(HOSTHDR)
	mkdir -p build/asan
	$(CXX) $(SANFLAGS) -o $@ $<
build/asan/test_planners: tests/cpp/test_planners.cpp cask_amd/csrc/plan_host.hpp cask_amd/csrc/plan_types.hpp cask_amd/csrc/trsv_lanes_plan.hpp $(HOSTHDR)
	mkdir -p build/asan
	$(CXX) $(SANFLAGS) -pthread -o $@ $<
build/asan/libcask_oracle.so: oracle/cask_oracle.c
	mkdir -p build/asan
	gcc -std=c11 -O1 -g -fno-omit-frame-pointer -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=all -fPIC -shared -o $@ $< -lm
# the threaded staging copy of the host-vector entry point: ASan / UBSan, and ThreadSanitizer in a build of its own
build/asan/test_host_copy: tests/cpp/test_host_copy.cpp cask_amd/csrc/host_copy.hpp
	mkdir -p build/asan
	$(CXX) $(SANFLAGS) -pthread -o $@ $<
build/asan/test_host_copy_tsan: tests/cpp/test_host_copy.cpp cask_amd/csrc/host_copy.hpp
	mkdir -p build/asan
	$(CXX) -std=c++17 -O1 -g -fsanitize=thread -Wall -Icask_amd/csrc -pthread -o $@ $<
asan: build/asan/test_host build/asan/ingest_time build/asan/test_planners build/asan/libcask_oracle.so build/asan/test_host_copy build/asan/test_host_copy_tsan
.PHONY: asan
