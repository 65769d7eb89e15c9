# Builds the MI355X device layer (HIP, gfx950), the C++ host library, the command-line
# driver and the CPU oracle.  hipcc cross-compiles without a GPU.
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
ARCH    ?= gfx950
LIBDIR  := logipathtracer_amd/lib
CSRC    := logipathtracer_amd/csrc
HOST    := logipathtracer_amd/host

# Bit-level parity with the CPU oracle needs: no FMA contraction, IEEE divide/sqrt, no fast-math.
# No SLP vectorisation: packing scalar f32 arithmetic into v_pk_* costs more v_mov shuffles than it
# saves (measured: 11 % on the persistent kernel); the one place where pairs pay is written by hand.
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC $(HIPFLAGS_EXTRA) -ffp-contract=off \
            -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wall -Wno-unused-function -Wno-pass-failed
CXXFLAGS := -O2 -std=c++17 -fPIC -ffp-contract=off -Wall -Wextra

HOST_SRCS := $(HOST)/lsg_min.cpp $(HOST)/png_decode.cpp $(HOST)/jpeg_decode.cpp $(HOST)/lsg_bvh.cpp $(HOST)/pt_scene_converter.cpp \
             $(HOST)/renderer_core.cpp $(HOST)/renderer_pt.cpp $(HOST)/host_capi.cpp
HOST_HDRS := $(wildcard $(HOST)/*.hpp) include/lpt.h
DEV_HDRS  := $(wildcard $(CSRC)/*.h) include/lpt.h

all: $(LIBDIR)/liblpt_hip.so $(LIBDIR)/liblpt_host.so $(LIBDIR)/lpt_render oracle

$(LIBDIR)/liblpt_hip.so: $(CSRC)/lpt_trace.h $(CSRC)/lpt_hip.hip $(CSRC)/lpt_lbvh.hip $(DEV_HDRS)
	mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/lpt_hip.hip $(CSRC)/lpt_lbvh.hip

$(LIBDIR)/liblpt_host.so: $(HOST_SRCS) $(HOST_HDRS) $(LIBDIR)/liblpt_hip.so
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRCS) -L$(LIBDIR) -llpt_hip -lz -pthread -Wl,-rpath,'$$ORIGIN'

$(LIBDIR)/lpt_render: $(HOST)/main.cpp $(LIBDIR)/liblpt_host.so
	$(CXX) $(CXXFLAGS) -o $@ $(HOST)/main.cpp -L$(LIBDIR) -llpt_host -llpt_hip -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIBDIR) oracle/build
.PHONY: all oracle clean
