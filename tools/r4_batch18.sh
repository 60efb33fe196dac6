#!/bin/bash
# round 4, batch 18: conv forward with 8-wave workgroups (one per CU) vs 4-wave ones; drug forward workgroup cap
for rep in 1 2; do
for W in 8 4; do
for F in 1024 16; do
  CGVP_CONV_FWD_WAVES=$W CGVP_GINE_FWD_WGS=$F python bench.py --no-cpu-baseline --epoch off --steps 300 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('conv_fwd_waves=$W gine_fwd_wgs=$F ms_per_step %.4f' % d['ms_per_step'])"
done
done
done
for W in 8 4; do
CGVP_CONV_FWD_WAVES=$W python bench.py --no-cpu-baseline --epoch off --steps 300 --only protein 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('conv_fwd_waves=$W protein only ms_per_step %.4f' % d['ms_per_step'])"
CGVP_CONV_FWD_WAVES=$W python bench.py --no-cpu-baseline --epoch off --steps 30 --workload long_graph_x64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('conv_fwd_waves=$W long_graph_x64 ms_per_step %.4f' % d['ms_per_step'])"
done
bash tools/trace_step.sh b18
