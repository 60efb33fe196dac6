#!/bin/bash
# round 4, batch 28: what the drug chain adds to the captured step when it has (almost) no work: queue / graph overhead vs CU sharing
for rep in 1 2; do
for args in "--only protein" "" "--drug-atoms 2" "--drug-atoms 10" "--drug-atoms 20"; do
  python bench.py --no-cpu-baseline --epoch off --steps 300 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('%-20s ms_per_step %.4f' % ('$args' or 'both', d['ms_per_step']))"
done
done
