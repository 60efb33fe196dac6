#!/bin/bash
# round 4, batch 17: protein backward workgroup cap across the bench configs
run() { python bench.py --no-cpu-baseline --epoch off "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('bwd_grid=$G %-50s ms_per_step %.4f' % ('$*', d['ms_per_step']))"; }
for G in 240 236 232 228; do
  export CGVP_BWD_GRID=$G
  run --steps 300
  run --steps 300 --workload kiba_b32
  run --steps 300 --workload bindingdb_b32_44 --dtype bf16
  run --steps 30 --workload long_graph_x64
done
