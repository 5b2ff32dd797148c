# Build of the MI355X trace path.  `make` builds everything that can be built on this host:
#   host    raytracer_amd/lib/librt_host.so   scene prep + reference-interface mirror (g++)
#   hip     raytracer_amd/lib/librt_hip.so    HIP kernels + rt_* C-ABI launcher (hipcc, gfx950)
#   oracle  oracle/librt_oracle.so            CPU restatement used only as the test checker (gcc)
#   ref     oracle/_ref/*                     llvmpipe runner of the reference shaders + glm pin
#                                             (only where /root/reference and Mesa exist)
ROOT    := $(abspath $(dir $(lastword $(MAKEFILE_LIST))))
LIBDIR  := $(ROOT)/raytracer_amd/lib
HOSTDIR := $(ROOT)/raytracer_amd/host
CSRC    := $(ROOT)/raytracer_amd/csrc
REFDIR  ?= /root/reference

CXX     ?= g++
CC      ?= gcc
HIPCC   ?= /opt/rocm/bin/hipcc

HOSTFLAGS := -std=c++17 -O2 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I$(ROOT)/include
# -fno-slp-vectorize: the SLP vectoriser pairs fp32 operations into v_pk_mul_f32 / v_pk_add_f32 and pays for every pair
# with register-pairing moves (extend kernel: 118 packed instructions, 206 moves; without it 0 and 115, the same
# instruction total, but the hot loops get shorter).  Measured on MI355X: metric frame 1.048 -> 0.94 ms, room.obj 38.0 ->
# 34.0 ms; the arithmetic is the same operation by operation (the GPU suite compares bit for bit).
HIPFLAGS  := --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function \
             -I$(ROOT)/include $(EXTRA_HIPFLAGS)
ORCFLAGS  := -std=c11 -O2 -fPIC -ffp-contract=off -fno-fast-math -fopenmp -Wall -Wno-unused-function \
             -I$(ROOT)/include

HOST_SRCS := $(HOSTDIR)/scene.cpp $(HOSTDIR)/png_write.cpp $(HOSTDIR)/host_cabi.cpp $(HOSTDIR)/render.cpp \
             $(HOSTDIR)/rt_loader.cpp $(HOSTDIR)/glfw_setter.cpp
HOST_HDRS := $(wildcard $(HOSTDIR)/*.hpp) $(wildcard $(ROOT)/include/*.h)
HIP_SRCS  := $(wildcard $(CSRC)/*.hip)
HIP_HDRS  := $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.hpp) $(wildcard $(ROOT)/include/*.h)

.PHONY: all host hip oracle ref clean headless debugwalk
all: host hip oracle headless

host: $(LIBDIR)/librt_host.so
hip: $(LIBDIR)/librt_hip.so
oracle: $(ROOT)/oracle/librt_oracle.so
headless: $(LIBDIR)/rt_headless

$(LIBDIR)/librt_host.so: $(HOST_SRCS) $(HOST_HDRS)
	@mkdir -p $(LIBDIR)
	$(CXX) $(HOSTFLAGS) -shared -o $@ $(HOST_SRCS) -ldl

$(LIBDIR)/rt_headless: $(HOSTDIR)/main.cpp $(LIBDIR)/librt_host.so
	$(CXX) $(HOSTFLAGS) -o $@ $(HOSTDIR)/main.cpp -L$(LIBDIR) -lrt_host -Wl,-rpath,'$$ORIGIN' -ldl -lpthread

# one object per translation unit (build/ is not tracked), so that an edit recompiles only its own file
OBJDIR   := $(ROOT)/build
HIP_OBJS := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(HIP_SRCS))
$(OBJDIR)/%.o: $(CSRC)/%.hip $(HIP_HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
$(LIBDIR)/librt_hip.so: $(HIP_OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(HIP_OBJS)

# Development build: every link / leaf range a walk follows is checked inside the kernels (trace3_device.h: RT_WALK_CHECK);
# run the GPU tests on it with RT_HIP_LIB=$(LIBDIR)/variants/librt_hip_debugwalk.so (tests/conftest.py reads the counters).
debugwalk: $(LIBDIR)/variants/librt_hip_debugwalk.so
$(LIBDIR)/variants/librt_hip_debugwalk.so: $(HIP_SRCS) $(HIP_HDRS)
	@mkdir -p $(LIBDIR)/variants
	$(HIPCC) $(HIPFLAGS) -DRT_DEBUG_WALK -shared -o $@ $(HIP_SRCS)

$(ROOT)/oracle/librt_oracle.so: $(ROOT)/oracle/rt_oracle.c $(ROOT)/oracle/rt_oracle.h
	$(CC) $(ORCFLAGS) -shared -o $@ $(ROOT)/oracle/rt_oracle.c -lm

# Built only in the container that holds the reference: nothing under oracle/_ref is committed.
ref:
	$(MAKE) -C $(ROOT)/oracle -f Makefile.ref REFDIR=$(REFDIR)

clean:
	rm -f $(LIBDIR)/*.so $(LIBDIR)/rt_headless $(ROOT)/oracle/librt_oracle.so
	rm -rf $(ROOT)/oracle/_ref $(OBJDIR)
