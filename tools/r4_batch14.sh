#!/bin/bash
for cfg in "1:0:" "0:0:" "0:0:--only drug" "0:0:--only protein" "0:1:" "0:1:--only drug"; do
  B=${cfg%%:*}; rest=${cfg#*:}; R=${rest%%:*}; args=${rest#*:}
  CGVP_BRIDGE=$B CGVP_GINE_RECOMPUTE=$R python bench.py --no-cpu-baseline --epoch off --steps 300 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('bridge=$B recompute=$R %-20s ms_per_step %.4f' % ('$args' or 'default', d['ms_per_step']))"
done
