#!/bin/bash
# A/B of run-time variants selected by ONE environment variable, in one GPU-box call, round-robin twice.
# Usage: bash tools/ab_env.sh VAR "v1 v2 ..." [bench.py args]      e.g.  bash tools/ab_env.sh CGVP_CONV_BWD "1 2" --workload long_graph_x64
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do
  for v in $VALS; do
    env $VAR=$v python bench.py --no-cpu-baseline --epoch off --steps 200 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d.get('roofline') or {}
print('%s=%-4s ms_per_step %.4f  pairs/s %.0f  dominant %s us' % ('$VAR', '$v', d['ms_per_step'], d['value'], str(r.get('avg_us')) + ' other ' + str(r.get('other'))))"
  done
done
