#!/bin/bash
# round 4, batch 16: how many CUs the protein backward leaves free x how many workgroups the drug forward takes
for rep in 1 2; do
for G in 240 232 224 208; do
for F in 1024 16 8; do
  CGVP_BWD_GRID=$G CGVP_GINE_FWD_WGS=$F python bench.py --no-cpu-baseline --epoch off --steps 300 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('bwd_grid=$G gine_fwd_wgs=$F ms_per_step %.4f' % d['ms_per_step'])"
done
done
done
CGVP_BWD_GRID=224 python bench.py --no-cpu-baseline --epoch off --steps 300 --only protein 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('bwd_grid=224 protein only ms_per_step %.4f' % d['ms_per_step'])"
