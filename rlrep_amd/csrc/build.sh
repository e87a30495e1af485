#!/bin/bash
# Build librlrep_hip.so for gfx950 (MI355X).  Cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
OBJ="$HERE/${OBJDIR:-.obj}"          # OBJDIR / OUTNAME: instrumented variants beside the product library (EXTRA_FLAGS=-DRL_TIMING_NC ...)
LIBNAME="${OUTNAME:-librlrep_hip.so}"
mkdir -p "$OUT" "$OBJ"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $EXTRA_FLAGS"
pids=()
for f in gemm16 gemm_lds noisecritic elementwise replearn rowprog xchain engine agents2 $EXTRA_SRCS; do
  if [ ! -f "$OBJ/$f.o" ] || [ "$HERE/$f.hip" -nt "$OBJ/$f.o" ] || [ -n "$(find "$HERE" -maxdepth 1 -name '*.h' -newer "$OBJ/$f.o")" ] || [ "$HERE/../../include/rlrep.h" -nt "$OBJ/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$OBJ/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/$LIBNAME" "$OBJ"/*.o
echo "built $OUT/$LIBNAME"
