#!/bin/bash
for rep in 1 2; do
for cfg in "0:" "1:" "0:--gine-bwd-wgs 8" "0:--gine-bwd-wgs 12" "0:--gine-bwd-wgs 20" "0:--gine-bwd-wgs 24" "0:--only drug" "1:--only drug"; do
  rc=${cfg%%:*}; args=${cfg#*:}
  CGVP_GINE_RECOMPUTE=$rc python bench.py --no-cpu-baseline --epoch off --steps 300 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('recompute=$rc %-28s ms_per_step %.4f' % ('$args' or 'default', d['ms_per_step']))"
done
done
