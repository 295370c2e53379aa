# Build of the product library (hipcc, gfx950) and of the test oracle (gcc).
# `python -c "import __graft_entry__ as g; g.build()"` drives this.
PKG      := cuda-pathtracer_amd
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# -ffp-contract=off: the kernel must execute the reference's IEEE op sequence
# -fno-slp-vectorize -fno-vectorize: packed f32 VALU ops issue at half rate on gfx950 and the packing costs v_mov
#   shuffles and spills; scalar code is 6.8 % faster (round 2; re-run: scripts/build_variants.sh + scripts/gpu_ab.sh)
HIPFLAGS := $(EXTRA_HIPFLAGS) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fno-vectorize -Iinclude -I$(PKG)/host -I$(PKG)/csrc \
            -Wall -Wextra -Wno-unused-parameter
LIB      := $(PKG)/libptamd.so
SRCS     := $(PKG)/csrc/pt_kernels.hip $(PKG)/csrc/pt_kernels_fma.hip $(PKG)/csrc/ptamd_api.cpp $(PKG)/host/scene_loader.cpp $(PKG)/host/bvh_builder.cpp $(PKG)/host/image_decode.cpp $(PKG)/host/image_png.cpp $(PKG)/host/image_resize.cpp
HDRS     := include/ptamd.h $(PKG)/host/ptamd_internal.h $(PKG)/csrc/pt_device.h $(PKG)/csrc/pt_launch.h
# identity of the build: EVERY source and header the library is compiled from + the flags, in this order (the host half decides what
# the device executes too: leaf size, launch geometry and slab sizing in ptamd_api.cpp, the tree's shape in bvh_builder.cpp).
# profiles/pmc_*.json carry the id of the build their counters were collected on; bench.py refuses to price a roofline with counters
# of another build.   Recompute: (cat $(SRCS) $(HDRS); echo "$(HIPFLAGS)") | sha256sum | cut -c1-16
BUILD_ID := $(shell (cat $(SRCS) $(HDRS); echo "$(HIPFLAGS)") | sha256sum | cut -c1-16)

ORACLE   := oracle/libpt_oracle.so

all: $(LIB) $(ORACLE)
lib: $(LIB)
oracle: $(ORACLE)

$(LIB): $(SRCS) $(HDRS)
	$(HIPCC) $(HIPFLAGS) -DPTAMD_BUILD_ID=\"$(BUILD_ID)\" -x hip -shared -o $@ $(SRCS)

$(ORACLE): oracle/pt_oracle.c oracle/pt_oracle.h
	gcc -O2 -std=c11 -ffp-contract=off -mfma -fPIC -shared -Wall -Wextra -o $@ oracle/pt_oracle.c -lm -lpthread

# oracle/_ref: the third-party libraries the reference vendors for the input side of the path (tinyobj,
# stb_image, stb_image_resize), compiled from where they lie under /root/reference.  Test infrastructure only;
# built only where the reference is present (the GPU box receives the prebuilt .so with the snapshot).
REFERENCE ?= /root/reference
REF3P_INC := $(REFERENCE)/cuda_opengl/3rd_party/include
REFLIB    := oracle/_ref/libref_thirdparty.so
ifneq ($(wildcard $(REF3P_INC)/tiny_obj_loader.h),)
oracle-ref: $(REFLIB)
$(REFLIB): oracle/ref_thirdparty.cpp
	@mkdir -p oracle/_ref
	g++ -O2 -std=c++11 -fPIC -shared -w -I$(REF3P_INC) -o $@ $<
else
oracle-ref:
	@echo "oracle-ref: $(REF3P_INC) not present, skipped"
endif

# optional OpenGL presenter (include/ptamd_gl.h): built only where GL headers and libGL exist; needs a display to RUN
GLLIB := $(PKG)/libptamd_gl.so
ifneq ($(wildcard /usr/include/GL/glext.h),)
gl: $(GLLIB)
$(GLLIB): $(PKG)/host/gl_presenter.cpp include/ptamd_gl.h include/ptamd.h
	$(HIPCC) -std=c++17 -O2 -fPIC -shared -Wall -Wextra -Iinclude $< -lGL -o $@
else
gl:
	@echo "gl: no GL headers on this host, skipped"
endif

clean:
	rm -f $(LIB) $(ORACLE) $(REFLIB) $(GLLIB)

.PHONY: all lib oracle oracle-ref gl clean san

# sanitizer harness of the HOST half (parsers, decoders, builder): g++ with ASan + UBSan over the same sources the library is
# built from; run by tests/test_sanitizers.py (GPU sanitizers are not available: the device half is covered by the parity suite)
SAN := build/host_san
HOST_SRCS := $(PKG)/host/scene_loader.cpp $(PKG)/host/bvh_builder.cpp $(PKG)/host/image_decode.cpp $(PKG)/host/image_png.cpp $(PKG)/host/image_resize.cpp
san: $(SAN)
$(SAN): tests/san/host_san.cpp $(HOST_SRCS) include/ptamd.h $(PKG)/host/ptamd_internal.h
	@mkdir -p build
	g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off -Wall -Wextra -Wno-unused-parameter \
	    -Iinclude -I$(PKG)/host -o $@ tests/san/host_san.cpp $(HOST_SRCS) -lpthread
