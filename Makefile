# Builds the MI355X-native graph-Laplacian filter: libglf.so (C-ABI, HIP gfx950),
# the image_processing host program and the CPU oracle used by the tests.
ROCM     ?= /opt/rocm
HIPCC    ?= $(ROCM)/bin/hipcc
ARCH     ?= gfx950
PKG      := image-processing-graph-laplacian_amd
CSRC     := $(PKG)/csrc
HOST     := $(PKG)/host
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-result -ffp-contract=fast -fno-slp-vectorize
CFLAGS   ?= -O2 -std=gnu11 -Wall -Wextra -fPIC

HIP_SRCS := $(CSRC)/ctx.hip $(CSRC)/affinity.hip $(CSRC)/eigen.hip $(CSRC)/nystroem.hip \
            $(CSRC)/filter.hip $(CSRC)/pipeline.hip $(CSRC)/comm.hip $(CSRC)/nlm.hip $(CSRC)/balance.hip
HIP_OBJS := $(HIP_SRCS:.hip=.o)
CPP_OBJS := $(CSRC)/host_util.o
C_OBJS   := $(HOST)/png_codec.o

all: $(PKG)/libglf.so $(PKG)/image_processing oracle

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/glf_internal.hpp $(CSRC)/nystroem_grid.inc $(CSRC)/nystroem_rank.inc $(CSRC)/nystroem_band.inc $(CSRC)/grid_common.inc $(CSRC)/affinity_grid.inc include/glf.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/host_util.o: $(CSRC)/host_util.cpp include/glf.h
	$(HIPCC) -O2 -std=c++17 -fPIC -Wall -c $< -o $@

$(HOST)/%.o: $(HOST)/%.c include/glf.h $(HOST)/stages.h
	gcc $(CFLAGS) -Iinclude -c $< -o $@

$(PKG)/libglf.so: $(HIP_OBJS) $(CPP_OBJS) $(C_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $^ -lz -pthread -ldl

$(PKG)/image_processing: $(HOST)/image_processing.o $(HOST)/stages.o $(PKG)/libglf.so
	gcc -o $@ $(HOST)/image_processing.o $(HOST)/stages.o -L$(PKG) -lglf -Wl,-rpath,'$$ORIGIN' -lm

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(CSRC)/*.o $(HOST)/*.o $(PKG)/libglf.so $(PKG)/image_processing
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
