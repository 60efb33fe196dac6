#!/bin/bash
O=gpurun_out
for args in "" "--issue-order drug" "--issue-order drug --gine-bwd-wgs 12" "--gine-bwd-wgs 24" "--only protein" "--only drug" "--drug-priority -1" "--issue-order drug --drug-priority -1"; do
  python bench.py --no-cpu-baseline --epoch off --steps 300 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('%-40s ms_per_step %.4f' % ('$args' or 'default', d['ms_per_step']))"
done
python -m pytest tests/test_hip_parity.py tests/test_custom_ops.py tests/test_hip_backward.py tests/test_bf16_storage.py -m gpu -q 2>&1 | tail -3
