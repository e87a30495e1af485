#!/bin/bash
# Build librlrep_hip.so for gfx950 (MI355X).  Cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/.obj"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $EXTRA_FLAGS"
pids=()
for f in gemm16 gemm_lds noisecritic elementwise replearn rowprog engine agents2 $EXTRA_SRCS; do
  if [ ! -f "$HERE/.obj/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/.obj/$f.o" ] || [ -n "$(find "$HERE" -maxdepth 1 -name '*.h' -newer "$HERE/.obj/$f.o")" ] || [ "$HERE/../../include/rlrep.h" -nt "$HERE/.obj/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/.obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/librlrep_hip.so" "$HERE"/.obj/*.o
echo "built $OUT/librlrep_hip.so"
