#!/bin/bash
for rep in 1 2; do
for cfg in "240:" "232:" "232:--gine-bwd-wgs 20" "232:--gine-bwd-wgs 24" "224:--gine-bwd-wgs 24" "224:--gine-bwd-wgs 32" "232:--only protein" "240:--only protein"; do
  G=${cfg%%:*}; args=${cfg#*:}
  CGVP_LIB_PATH=$PWD/caster-dta_amd/lib/ab/libcaster_gvp_g$G.so python bench.py --no-cpu-baseline --epoch off --steps 300 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('grid=$G %-28s ms_per_step %.4f' % ('$args' or 'default', d['ms_per_step']))"
done
done
