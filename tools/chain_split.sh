#!/bin/bash
# Which chain sets the captured encoders step: protein alone, drug alone, both (unprofiled HIP-graph replays), and the
# effect of the GINE backward workgroup cap.  Usage (GPU box): bash tools/chain_split.sh
for args in "--only protein" "--only drug" "" "--gine-bwd-wgs 8" "--gine-bwd-wgs 24" "--gine-bwd-wgs 32" "--collate-csr"; do
  python bench.py --no-cpu-baseline --epoch off --steps 300 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('%-22s ms_per_step %.4f' % ('$args' or 'both', d['ms_per_step']))"
done
