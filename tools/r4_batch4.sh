#!/bin/bash
O=gpurun_out
for cfg in "train 6 100" "eval 6 100" "eval 6 400" "train 64 300" "eval 64 300"; do
  timeout -k 10 200 python -X faulthandler tools/debug_graphed.py $cfg > $O/r4_dbg_graphed.log 2>&1; echo "graphed [$cfg] rc=$? : $(grep -E '^ok|^start' $O/r4_dbg_graphed.log | tr '\n' ' ')"
done
python -m pytest tests/test_hip_models.py -m gpu -q -k "leaf_grads" 2>&1 | tail -3
for v in 0 1; do
  CGVP_EXACT_LEAVES=$v python bench.py --steps 20 --epoch nominal --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); e=d['config']['epoch']
print('EXACT_LEAVES=$v eager %.4f host %.4f fast_passes %s fused %.4f joint eager %.3f' % (e['ms_per_step'], e['host_issue_ms_per_step'], e.get('eager_backward_passes_without_leaf_tasks'), e['eager_fused_parameters']['ms_per_step'], e['joint']['eager']['ms_per_step']))"
done
CGVP_BRIDGE_TIMING=1 python tools/host_profile_encoders.py 2>&1 | tail -30
