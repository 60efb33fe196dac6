#!/bin/bash
# round 4, batch 15: merged head+node and edge+embed launches, multi-workgroup CSR scan -- tests, then A/B
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_b15.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/gpu_tests_b15.log
for rep in 1 2; do
for cfg in "0:0:" "1:0:" "0:1:" "1:1:" "0:0:--only protein" "1:1:--only protein"; do
  H=${cfg%%:*}; rest=${cfg#*:}; T=${rest%%:*}; args=${rest#*:}
  CGVP_SPLIT_HEAD_BWD=$H CGVP_SPLIT_TAIL_BWD=$T python bench.py --no-cpu-baseline --epoch off --steps 300 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('split_head=$H split_tail=$T %-20s ms_per_step %.4f' % ('$args' or 'default', d['ms_per_step']))"
done
done
