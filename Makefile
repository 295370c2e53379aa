# Build of the product library (hipcc, gfx950) and of the test oracle (gcc).
# `python -c "import __graft_entry__ as g; g.build()"` drives this.
PKG      := cuda-pathtracer_amd
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# -ffp-contract=off: the kernel must execute the reference's IEEE op sequence
HIPFLAGS := $(EXTRA_HIPFLAGS) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$(PKG)/host -I$(PKG)/csrc \
            -Wall -Wextra -Wno-unused-parameter
LIB      := $(PKG)/libptamd.so
SRCS     := $(PKG)/csrc/pt_kernels.hip $(PKG)/csrc/ptamd_api.cpp $(PKG)/host/scene_loader.cpp $(PKG)/host/bvh_builder.cpp
HDRS     := include/ptamd.h $(PKG)/host/ptamd_internal.h $(PKG)/csrc/pt_device.h $(PKG)/csrc/pt_launch.h

ORACLE   := oracle/libpt_oracle.so

all: $(LIB) $(ORACLE)
lib: $(LIB)
oracle: $(ORACLE)

$(LIB): $(SRCS) $(HDRS)
	$(HIPCC) $(HIPFLAGS) -x hip -shared -o $@ $(SRCS)

$(ORACLE): oracle/pt_oracle.c oracle/pt_oracle.h
	gcc -O2 -std=c11 -ffp-contract=off -mfma -fPIC -shared -Wall -Wextra -o $@ oracle/pt_oracle.c -lm -lpthread

clean:
	rm -f $(LIB) $(ORACLE)

.PHONY: all lib oracle clean
