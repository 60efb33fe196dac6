#!/bin/bash
# protein alone / drug alone / both, unprofiled replays (short form of chain_split.sh)
for args in "--only protein" "--only drug" "" ""; do
  python bench.py --no-cpu-baseline --epoch off --steps 300 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('%-22s ms_per_step %.4f' % ('$args' or 'both', d['ms_per_step']))"
done
