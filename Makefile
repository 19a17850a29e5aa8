# Builds libmtc.so (HIP kernels + C ABI) for gfx950, in-tree.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := matryoshka_transcoder_amd/csrc
SRCS  := $(CSRC)/api.hip $(CSRC)/encode.hip $(CSRC)/select.hip $(CSRC)/decode.hip $(CSRC)/backward.hip $(CSRC)/optim.hip $(CSRC)/aux.hip $(CSRC)/collect.hip $(CSRC)/store.hip $(CSRC)/comm.hip
OBJS  := $(SRCS:.hip=.o)
LIB   := matryoshka_transcoder_amd/libmtc.so
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result

all: $(LIB)

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/common.hpp $(CSRC)/rowops.hpp include/mtc.h
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl

clean:
	rm -f $(OBJS) $(LIB)

.PHONY: all clean
