#!/bin/bash
# A/B library builds: recompile ONE translation unit with extra -D flags and link it with the regular objects.
# Usage: bash tools/build_variant.sh <tag> <unit> "<flags>"   -> caster-dta_amd/lib/ab/libcaster_gvp_<tag>.so
# Run a bench against it with CGVP_LIB_PATH=<that file> (the Python custom-op host path is used then).
set -e
TAG=$1; UNIT=$2; FLAGS=$3
HERE=caster-dta_amd/csrc; OBJ=caster-dta_amd/lib/_obj; OUT=caster-dta_amd/lib/ab
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o $OUT/${UNIT}_$TAG.o $HERE/$UNIT.hip 2> $OUT/${UNIT}_$TAG.log
OBJS=""
for s in gvp_kernels gvp_quad_kernels gvp_quad_bwd_kernels gine_quad_kernels pass_api attn_kernels feat_kernels linear_kernels norm_kernels elementwise_kernels; do
  if [ "$s" == "$UNIT" ]; then OBJS="$OBJS $OUT/${UNIT}_$TAG.o"; else OBJS="$OBJS $OBJ/$s.o"; fi
done
hipcc --offload-arch=gfx950 -fPIC -shared -o $OUT/libcaster_gvp_$TAG.so $OBJS
rm -f $OUT/${UNIT}_$TAG.o
ls -la $OUT/libcaster_gvp_$TAG.so
