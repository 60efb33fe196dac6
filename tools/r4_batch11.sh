#!/bin/bash
O=gpurun_out
python -X faulthandler -m pytest tests -m gpu -q > $O/r4_gpu10.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/r4_gpu10.log; grep -E "^FAILED|passed|failed|^E  |Fatal" $O/r4_gpu10.log | cut -c1-300 | head -20
for args in "" "--only drug" "--only protein" "--workload kiba_b32" "--config 5"; do
  python bench.py --no-cpu-baseline --epoch off --steps 300 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('%-40s ms_per_step %.4f' % ('$args' or 'default', d['ms_per_step']))"
done
python bench.py --scope joint --steps 50 --no-cpu-baseline --epoch off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('joint graph ms_per_step %.4f' % d['ms_per_step'])"
exit $rc
