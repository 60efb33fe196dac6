#!/bin/bash
# diagnostic sweeps of bench.py flags: prints ms_per_step per setting
run() { python bench.py --no-cpu-baseline --epoch off "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; }
for i in 1 2; do run --drug-priority 0; run --drug-priority -1; done
for w in 8 12 16 24 32; do run --gine-bwd-wgs $w; done
