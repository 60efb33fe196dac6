#!/bin/bash
# A/B of library variants built by tools/build_variant.sh IN ONE GPU-box call (timings of different calls land on
# different boxes and differ by +-3 us): for each tag, the captured encoders step with both encoders / the drug alone.
# Usage: bash tools/ab_libs.sh <tag> [<tag> ...]     (twice round-robin, to see the run-to-run spread)
for rep in 1 2; do
for tag in "$@"; do
  for args in "" "--only drug"; do
    CGVP_LIB_PATH=$PWD/caster-dta_amd/lib/ab/libcaster_gvp_$tag.so python bench.py --no-cpu-baseline --epoch off --steps 300 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('%-10s %-12s ms_per_step %.4f' % ('$tag', '$args' or 'both', d['ms_per_step']))"
  done
done
done
