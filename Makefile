# Builds librag_amd.so (HIP kernels + C ABI) for gfx950, in-tree.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := rag_amd/csrc
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(SRCS:.hip=.o)
LIB   := rag_amd/lib/librag_amd.so
CXXFLAGS := -O3 -std=c++20 -fPIC --offload-arch=$(ARCH) -Iinclude -I$(CSRC) -Wall -Wno-unused-function $(EXTRA)
# `make clean && make DIAG=1` builds the profiling library: it honours the RAGMI_K3_DIAG_NOSTORE / RAGMI_K3_WLDS switches that
# skip stores / the MFMA block / the staging of conv3d_k3 (profiles/README.md).  The default build contains none of them.
ifeq ($(DIAG),1)
CXXFLAGS += -DRAGMI_DIAG
endif

all: $(LIB)

# Per-file code generation choices, each measured on the MI355X with both builds on the same box (tools/ab_bench.sh, rocprofv3):
# the SLP vectorizer packs the soft-argmin's independent fp32 chains into v_pk_* pairs and pays for it in v_mov shuffles
# (57 -> 47 us with it off); the cost-volume planes kernel gains from the same packing (54 us with, 60 without), so it stays on elsewhere.
$(CSRC)/disp.o: CXXFLAGS += -fno-slp-vectorize
# same for the single-output-channel convolution: 434 v_pk_fma_f32 + 160 v_mov_b32 instead of 864 v_fmac_f32 (packed fp32 issues at the same FLOP rate)
$(CSRC)/conv3d_c1.o: CXXFLAGS += -fno-slp-vectorize

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/conv3d_k3.h $(CSRC)/conv3d_x3_common.h include/rag_amd.h
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p rag_amd/lib
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS)

clean:
	rm -f $(OBJS) $(LIB)

.PHONY: all clean
