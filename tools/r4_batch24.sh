#!/bin/bash
# round 4, batch 24: store-once weight-gradient accumulation for single-tile node / head / embed launches
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_b24.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/gpu_tests_b24.log
for rep in 1 2 3; do
for O in 1 0; do
for args in "" "--only protein"; do
  CGVP_BWD_STORE_ONCE=$O python bench.py --no-cpu-baseline --epoch off --steps 300 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('store_once=$O %-20s ms_per_step %.4f' % ('$args' or 'default', d['ms_per_step']))"
done
done
done
