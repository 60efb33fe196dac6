#!/bin/bash
# Build libcaster_gvp.so for gfx950 (cross-compiles without a GPU).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
TMP="$HERE/../lib/_obj"
mkdir -p "$OUT" "$TMP"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
  -o "$TMP/libcaster_gvp.so" "$HERE/gvp_kernels.hip" "$HERE/gvp_quad_kernels.hip" "$HERE/gvp_quad_bwd_kernels.hip" "$HERE/gine_quad_kernels.hip" \
  -Rpass-analysis=kernel-resource-usage -save-temps=obj 2> "$TMP/resource_usage.txt" || { cat "$TMP/resource_usage.txt"; exit 1; }
mv "$TMP/libcaster_gvp.so" "$OUT/libcaster_gvp.so"
grep -E "error|warning:" "$TMP/resource_usage.txt" || true
