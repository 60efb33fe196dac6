#!/bin/bash
# Build libcaster_gvp.so for gfx950 (cross-compiles without a GPU).  The translation units are
# compiled in parallel (one hipcc per .hip file), then linked.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
TMP="$HERE/../lib/_obj"
mkdir -p "$OUT" "$TMP"
SRCS=(gvp_kernels gvp_quad_kernels gvp_quad_bwd_kernels gine_quad_kernels pass_api)
[ -f "$HERE/attn_kernels.hip" ] && SRCS+=(attn_kernels)
[ -f "$HERE/feat_kernels.hip" ] && SRCS+=(feat_kernels)
[ -f "$HERE/linear_kernels.hip" ] && SRCS+=(linear_kernels)
[ -f "$HERE/norm_kernels.hip" ] && SRCS+=(norm_kernels)
[ -f "$HERE/elementwise_kernels.hip" ] && SRCS+=(elementwise_kernels)
pids=()
for s in "${SRCS[@]}"; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -o "$TMP/$s.o" "$HERE/$s.hip" \
      -Rpass-analysis=kernel-resource-usage -save-temps=obj 2> "$TMP/$s.log" ) &
  pids+=($!)
done
fail=0
for i in "${!pids[@]}"; do
  if ! wait "${pids[$i]}"; then fail=1; echo "== ${SRCS[$i]}.hip failed"; grep -E "error|Error" -A3 "$TMP/${SRCS[$i]}.log" | head -60; fi
done
[ $fail -eq 0 ] || exit 1
: > "$TMP/resource_usage.txt"
for s in "${SRCS[@]}"; do cat "$TMP/$s.log" >> "$TMP/resource_usage.txt"; done
hipcc --offload-arch=gfx950 -fPIC -shared -o "$TMP/libcaster_gvp.so" $(for s in "${SRCS[@]}"; do echo "$TMP/$s.o"; done)
mv "$TMP/libcaster_gvp.so" "$OUT/libcaster_gvp.so"
grep -E "error|warning:" "$TMP/resource_usage.txt" || true
