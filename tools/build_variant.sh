#!/bin/bash
# tools/build_variant.sh NAME FILE.hip [-DMACRO ...]: a copy of libmtc.so with ONE source rebuilt under extra macros
# (A/B timing of kernel variants through MTC_LIB; the product library is untouched).  Run on the build container.
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
CS=matryoshka_transcoder_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $CS/$SRC.hip -o /tmp/${SRC}_$NAME.o
OBJS=""
for f in api encode select decode backward optim aux collect store comm; do
  if [ "$f" = "$SRC" ]; then OBJS="$OBJS /tmp/${SRC}_$NAME.o"; else OBJS="$OBJS $CS/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o matryoshka_transcoder_amd/libmtc_exp_$NAME.so $OBJS -ldl
echo built matryoshka_transcoder_amd/libmtc_exp_$NAME.so
