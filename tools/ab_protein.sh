#!/bin/bash
# A/B of library variants (tools/build_variant.sh) in one GPU-box call: protein chain alone and the full step.
for rep in 1 2; do
for tag in "$@"; do
  for args in "--only protein" ""; do
    CGVP_LIB_PATH=$PWD/caster-dta_amd/lib/ab/libcaster_gvp_$tag.so python bench.py --no-cpu-baseline --epoch off --steps 300 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d.get('roofline') or {}; print('%-10s %-15s ms_per_step %.4f  conv_bwd %s us' % ('$tag', '$args' or 'both', d['ms_per_step'], r.get('avg_us')))"
  done
done
done
