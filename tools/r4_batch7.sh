#!/bin/bash
O=gpurun_out
for cfg in "eval 6 100 fwd" "train 6 100 backward" "train 6 100 none"; do
  timeout -k 10 200 python -X faulthandler tools/debug_graphed.py $cfg > $O/r4_dbg_graphed.log 2>&1; echo "graphed [$cfg] rc=$? : $(grep -E '^ok|^start|RuntimeError' $O/r4_dbg_graphed.log | cut -c1-150 | tr '\n' ' ')"
done
python -X faulthandler -m pytest tests -m gpu -q > $O/r4_gpu8.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/r4_gpu8.log; grep -E "^FAILED|passed|failed|^E  |Fatal" $O/r4_gpu8.log | cut -c1-300 | head -30
CGVP_BRIDGE_TIMING=1 python tools/host_profile_encoders.py > $O/r4_host_profile.txt 2>&1; grep -E "host issue" $O/r4_host_profile.txt; tail -8 $O/r4_host_profile.txt
python bench.py --steps 20 --epoch nominal --no-cpu-baseline 2> $O/r4_bench3.err | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); e=d['config']['epoch']
print('eager %.4f host %.4f fast_passes %s fused %.4f bucketed %.4f joint eager %.3f bucketed %.3f' % (e['ms_per_step'], e['host_issue_ms_per_step'], e.get('eager_backward_passes_without_leaf_tasks'), e['eager_fused_parameters']['ms_per_step'], e['bucketed_graphs']['ms_per_step'], e['joint']['eager']['ms_per_step'], e['joint']['bucketed_graphs']['ms_per_step']))"; tail -3 $O/r4_bench3.err
exit $rc
