# Builds the product library (HIP, gfx950) and, for the tests, the CPU oracle.
#   make            -> dxrexperiments_amd/lib/libdxrexperiments_amd.so
#   make oracle     -> oracle/liboracle.so   (test infrastructure only)
# -ffp-contract=off is REQUIRED: the kernels' arithmetic is defined without FMA
# (bit parity with the oracle); divide and sqrt must be the correctly rounded forms.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC = dxrexperiments_amd/csrc
LIBDIR = dxrexperiments_amd/lib
LIBNAME ?= libdxrexperiments_amd.so
LIB = $(LIBDIR)/$(LIBNAME)
HIPFLAGS = -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -fno-fast-math \
           -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize \
           -Wall -Wno-unused-function -Iinclude $(EXTRA)
SRCS = $(CSRC)/rt_api.hip $(CSRC)/rt_bvh_build.hip $(CSRC)/rt_bvh_ploc.hip $(CSRC)/rt_bvh_wide.hip $(CSRC)/rt_trace.hip $(CSRC)/rt_pipeline.hip $(CSRC)/rt_pipeline_render.hip $(CSRC)/rt_pipeline_host.hip $(CSRC)/rt_denoise.hip $(CSRC)/rt_dist.hip \
       $(CSRC)/rt_obj.cpp $(CSRC)/rt_fbx.cpp $(CSRC)/rt_host.cpp $(CSRC)/rt_dds.cpp $(CSRC)/rt_image.cpp
HDRS = $(wildcard $(CSRC)/*.h) include/dxr_amd.h include/dxr_amd_types.h
OBJS = $(patsubst $(CSRC)/%,build/%.o,$(SRCS))

BIN = $(LIBDIR)/progressive $(LIBDIR)/realtime_denoise $(LIBDIR)/test_wrapper $(LIBDIR)/progressive_multi
CXX ?= g++
HOSTFLAGS = -O2 -std=c++17 -Wall -Iinclude -Idxrexperiments_amd/include
HOSTLINK = -L$(LIBDIR) -ldxrexperiments_amd -L/opt/rocm/lib -Wl,-rpath,'$$ORIGIN' -Wl,-rpath-link,/opt/rocm/lib

all: $(LIB) $(BIN) build/rt_rccl_abi_check.o

# the hand-declared RCCL prototypes (rt_rccl_abi.h; RCCL is dlopen'ed) against the installed <rccl/rccl.h>: compiled, linked into nothing
build/rt_rccl_abi_check.o: $(CSRC)/rt_rccl_abi_check.cpp $(CSRC)/rt_rccl_abi.h
	@mkdir -p build
	$(HIPCC) -std=c++17 --offload-arch=$(ARCH) -x hip -I/opt/rocm/include -c $< -o $@

# host programs written against the reference-shaped C++ API (no HIP needed to compile them)
$(LIBDIR)/progressive: examples/progressive.cpp $(LIB) $(wildcard dxrexperiments_amd/include/*.h)
	$(CXX) $(HOSTFLAGS) $< -o $@ $(HOSTLINK)

$(LIBDIR)/realtime_denoise: examples/realtime_denoise.cpp $(LIB) $(wildcard dxrexperiments_amd/include/*.h)
	$(CXX) $(HOSTFLAGS) $< -o $@ $(HOSTLINK)

$(LIBDIR)/progressive_multi: examples/progressive_multi.cpp $(LIB) $(wildcard dxrexperiments_amd/include/*.h)
	$(CXX) $(HOSTFLAGS) $< -o $@ $(HOSTLINK)

$(LIBDIR)/test_wrapper: tests/cpp/test_wrapper.cpp $(LIB) $(wildcard dxrexperiments_amd/include/*.h)
	$(CXX) $(HOSTFLAGS) $< -o $@ $(HOSTLINK)

build/%.hip.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

build/%.cpp.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS) -ldl -lz

oracle:
	$(MAKE) -C oracle liboracle.so

clean:
	rm -rf build $(LIB) $(BIN)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
