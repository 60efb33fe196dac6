#!/bin/bash
O=gpurun_out
export TMPDIR=/tmp
for args in "" "--only drug" "--workload kiba_b32"; do
  python bench.py --no-cpu-baseline --epoch off --steps 300 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('%-40s ms_per_step %.4f' % ('$args' or 'default', d['ms_per_step']))"
done
python -m pytest tests/test_hip_parity.py tests/test_hip_random_graphs.py tests/test_gine_depth4.py -m gpu -q 2>&1 | tail -2
mkdir -p $O/trace_scan
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_scan -o run -- python3 bench.py --no-cpu-baseline --epoch off --steps 60 > $O/trace_scan.log 2>&1
python tools/step_trace.py $O/trace_scan > $O/r4_step_trace_scan.txt 2>/dev/null; cat $O/r4_step_trace_scan.txt | cut -c1-150
rm -rf $O/trace_scan
