#!/bin/bash
# Build librlrep_hip.so for gfx950 (MI355X).  Cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
# RLREP_BUILD_EXPERIMENTS=1: the opt-in engines that were measured and NOT adopted (row-block programs rowprog.hip, per-XCD persistent
# chains xchain.hip, first layers fused into the second, superseded noise-critic forward kernels) are compiled in as well, into a library
# of their own (librlrep_hip_exp.so; RLREP_LIB selects it).  The product library carries stubs that report them as not built.
if [ -n "$RLREP_BUILD_EXPERIMENTS" ]; then
  EXTRA_FLAGS="$EXTRA_FLAGS -DRL_EXPERIMENTS"; EXP_SRCS="rowprog xchain"; OBJDIR="${OBJDIR:-.obj_exp}"; OUTNAME="${OUTNAME:-librlrep_hip_exp.so}"
else
  EXP_SRCS="experiments_off"
fi
OBJ="$HERE/${OBJDIR:-.obj}"          # OBJDIR / OUTNAME: instrumented variants beside the product library (EXTRA_FLAGS=-DRL_TIMING_NC ...)
LIBNAME="${OUTNAME:-librlrep_hip.so}"
mkdir -p "$OUT" "$OBJ"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $EXTRA_FLAGS"
pids=()
for f in gemm16 gemm_lds noisecritic elementwise replearn comm $EXP_SRCS engine agents2 $EXTRA_SRCS; do
  if [ ! -f "$OBJ/$f.o" ] || [ "$HERE/$f.hip" -nt "$OBJ/$f.o" ] || [ -n "$(find "$HERE" -maxdepth 1 -name '*.h' -newer "$OBJ/$f.o")" ] || [ "$HERE/../../include/rlrep.h" -nt "$OBJ/$f.o" ]; then
    PF=""; { [ "$f" = gemm16 ] || [ "$f" = elementwise ] || [ "$f" = gemm_lds ]; } && PF="-mllvm -amdgpu-kernarg-preload-count=14"     # gemm16_kernel / adam_kernel / the gemm_lds kernels: leading scalars preloaded into SGPRs
    $HIPCC $FLAGS $PF -c "$HERE/$f.hip" -o "$OBJ/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
LINK=""; for f in gemm16 gemm_lds noisecritic elementwise replearn comm $EXP_SRCS engine agents2 $EXTRA_SRCS; do LINK="$LINK $OBJ/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/$LIBNAME" $LINK
echo "built $OUT/$LIBNAME"
