#!/bin/bash
# Builds the diagnostic library (stamps compiled in) next to the product one; run on the build container.
set -e
cd "$(dirname "$0")/.."
CS=matryoshka_transcoder_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS -DMTC_ENCODE_STAMPS -c $CS/encode.hip -o /tmp/encode_diag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o matryoshka_transcoder_amd/libmtc_diag.so /tmp/encode_diag.o $CS/api.o $CS/select.o $CS/decode.o $CS/backward.o $CS/optim.o $CS/aux.o $CS/collect.o $CS/store.o $CS/comm.o -ldl
echo built matryoshka_transcoder_amd/libmtc_diag.so
