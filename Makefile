# Builds, in-tree:
#   volume-renderer_amd/lib/libvolly_gpu.so    HIP kernels + C-ABI (gfx950)
#   volume-renderer_amd/lib/libvolly_host.so   host scene pipeline (C++17)
#   oracle/libvolly_oracle.so                  CPU oracle (test infrastructure)
#   oracle/_ref/libtricube_ref.so              reference triangleCube.c, only when /root/reference exists
# Float semantics: the reference is built without FMA and without fast-math
# (reference Makefile:3), so every target uses -ffp-contract=off.

PKG      := volume-renderer_amd
LIBDIR   := $(PKG)/lib
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
CC       ?= gcc
REF      ?= /root/reference

HOST_SRC := $(PKG)/host/mesh.cpp $(PKG)/host/voxelize.cpp $(PKG)/host/tricube.cpp \
            $(PKG)/host/svo.cpp $(PKG)/host/camera.cpp $(PKG)/host/volly_host_c.cpp
HOST_HDR := $(PKG)/host/volly_scene.h $(PKG)/host/vollymath.h $(PKG)/host/tricube.h include/volly_host.h include/volly_gpu.h

FPFLAGS  := -ffp-contract=off -fno-fast-math
# the SLP vectorizer pairs the DDA's x/y float operations into v_pk_*_f32, which issue at half
# rate on gfx950 and cost extra moves: measured 4-5 % slower (profiles/ab_bench.py, variant "slp")
GPUFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $(FPFLAGS) -fno-slp-vectorize -Iinclude

all: gpu host oracle ref examples tools

gpu: $(LIBDIR)/libvolly_gpu.so
host: $(LIBDIR)/libvolly_host.so
oracle: oracle/libvolly_oracle.so

GPU_SRC := $(PKG)/csrc/volly_gpu.hip $(PKG)/csrc/volly_voxelize.hip
GPU_HDR := $(PKG)/csrc/volly_kernels.h $(PKG)/csrc/volly_internal.h include/volly_gpu.h tools/check_kernel_contract.py

# What the library was built from, recorded inside it (volly_gpu_build_info): the SHA-256 of its sources in the order
# of GPU_HASHED, and the compiler whose output passed the kernel contract.  tests/test_abi.py recomputes the hash from
# the tree, so a prebuilt .so that no longer matches its sources -- the only way the assembly loops' register
# contract could go unchecked -- fails the CPU suite here and on the GPU box.
GPU_HASHED := $(GPU_SRC) $(PKG)/csrc/volly_kernels.h $(PKG)/csrc/volly_internal.h include/volly_gpu.h
GPU_SRC_SHA = $(shell cat $(GPU_HASHED) | sha256sum | cut -c1-16)
HIPCC_VER = $(shell $(HIPCC) --version | grep -m1 -i 'HIP version' | tr -d '"' | sed 's/  */ /g')
BUILDINFO = -DVOLLY_BUILD_SRC_SHA='"$(GPU_SRC_SHA)"' -DVOLLY_BUILD_HIPCC='"$(HIPCC_VER)"'

$(LIBDIR)/libvolly_gpu.so: $(GPU_SRC) $(GPU_HDR)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(GPUFLAGS) $(BUILDINFO) -o $@ $(GPU_SRC)
	python3 tools/check_kernel_contract.py || { rm -f $@; exit 1; }

$(LIBDIR)/libvolly_host.so: $(HOST_SRC) $(HOST_HDR)
	@mkdir -p $(LIBDIR)
	$(CXX) -O2 -std=c++17 -fPIC -shared -pthread $(FPFLAGS) -Wall -Wextra -o $@ $(HOST_SRC)

oracle/libvolly_oracle.so: oracle/volly_oracle.c oracle/volly_oracle.h include/volly_gpu.h
	$(CC) -O3 -mavx -std=gnu11 -fPIC -shared -fopenmp $(FPFLAGS) -Wall -Wextra -o $@ oracle/volly_oracle.c -lm

# the one reference source that compiles stand-alone (as C++: it includes <cmath>)
ref:
	@if [ -f $(REF)/triangleCube.c ]; then $(MAKE) oracle/_ref/libtricube_ref.so; \
	 else echo "no $(REF): keeping prebuilt oracle/_ref (if any)"; fi

oracle/_ref/libtricube_ref.so: oracle/tricube_ref_export.cpp $(REF)/triangleCube.c
	@mkdir -p oracle/_ref
	$(CXX) -O3 -mavx -std=c++14 -fPIC -shared $(FPFLAGS) -I$(REF) -o $@ oracle/tricube_ref_export.cpp

examples: examples/volly_headless

# device-side exhaustive check of the short division / sqrt sequences the kernels use
tools: tools/exhaustive_math tools/valu_rate
tools/exhaustive_math: tools/exhaustive_math.hip
	$(HIPCC) --offload-arch=gfx950 -O3 $(FPFLAGS) -Wno-unused-result -o $@ $<
# issue rate of the instruction kinds the DDA loop is made of
tools/valu_rate: tools/valu_rate.hip
	$(HIPCC) --offload-arch=gfx950 -O3 -Wno-unused-result -o $@ $<
# floor(p) + h and the packed cell from p + 1.5 * 2^23 rounded toward -infinity: exactness and step cost
tools/_bin/magic_floor: tools/magic_floor.hip
	@mkdir -p tools/_bin
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 $(FPFLAGS) -Wno-unused-result -o $@ $<

# alternative builds of the kernels for A/B measurement (profiles/ab_bench.py); never loaded by the product
VARDIR := $(LIBDIR)/variants
VARIANTS ?= checked:-DVK_FAST_DDA=0 occ1:-DVK_OCC_LEVELS=1 rootwalk:-DVK_SVO_TABLE=0 waves4:-DVK_WAVE_BLOCK=4 \
            longdiv:-DVK_SHORT_DIV=0 ckpt4:-DVK_CKPT_STRIDE=4 noasm:-DVK_ASM_QUIET=0 \
            noasmextra:-DVK_ASM_EXTRA=0 noasmbricks:-DVK_ASM_BRICKS=0 svocpp:-DVK_ASM_SVO=0 svonodes:-DVK_SVO_BLOCKS=0 \
            slp:-fslp-vectorize ahead0:-DVK_RUN_AHEAD=0 count:-DVK_COUNT=1
variants:
	@mkdir -p $(VARDIR)
	for v in $(VARIANTS); do \
	  n=$${v%%:*}; f="$${v#*:}"; \
	  $(HIPCC) $(GPUFLAGS) $(BUILDINFO) $$f -o $(VARDIR)/libvolly_gpu_$$n.so $(GPU_SRC) || exit 1; \
	  python3 tools/check_kernel_contract.py $$f || exit 1; \
	done

# reference-style main loop on top of the C-ABI (links both libraries; rpath keeps it relocatable with the tree)
examples/volly_headless: examples/volly_headless.cpp $(PKG)/host/volly_state.h $(LIBDIR)/libvolly_host.so $(LIBDIR)/libvolly_gpu.so
	$(CXX) -O2 -std=c++17 $(FPFLAGS) -Wall -Wextra -o $@ examples/volly_headless.cpp \
	    -L$(LIBDIR) -lvolly_host -lvolly_gpu -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath-link,/opt/rocm/lib

# Host side of both libraries under AddressSanitizer (clang's runtime; device code is left
# alone: GPU sanitizers are not available on the pool) plus a driver over the whole C-ABI.
# Run build/san/gpu_sanitize_driver assets on a box with an MI355X.
SANDIR := build/san
CLANGXX ?= /opt/rocm/lib/llvm/bin/clang++
sanitize-gpu:
	@mkdir -p $(SANDIR)/lib
	$(HIPCC) --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared $(FPFLAGS) -fno-slp-vectorize -Xarch_host -fsanitize=address \
	    -Xarch_host -fno-omit-frame-pointer -Iinclude -o $(SANDIR)/lib/libvolly_gpu.so $(GPU_SRC)
	$(CLANGXX) -O1 -g -std=c++17 -fPIC -shared -pthread $(FPFLAGS) -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -Iinclude \
	    -o $(SANDIR)/lib/libvolly_host.so $(HOST_SRC)
	$(CLANGXX) -O1 -g -std=c++17 $(FPFLAGS) -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -Iinclude -I/opt/rocm/include \
	    -D__HIP_PLATFORM_AMD__ -o $(SANDIR)/gpu_sanitize_driver tests/gpu_sanitize_driver.cpp \
	    -L$(SANDIR)/lib -lvolly_host -lvolly_gpu -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/lib' -Wl,-rpath,/opt/rocm/lib

clean:
	rm -rf $(LIBDIR) oracle/libvolly_oracle.so oracle/_ref examples/volly_headless tools/exhaustive_math tools/valu_rate

.PHONY: sanitize-gpu all gpu host oracle ref examples tools variants clean
