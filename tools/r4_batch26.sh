#!/bin/bash
# round 4, batch 26: the joint head's row-wise ops through the C++ bridge in eager mode
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_b26.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/gpu_tests_b26.log
python tools/host_profile_joint.py 2>&1 | grep -E "host issue|TRAIN step"
CGVP_BRIDGE=0 python tools/host_profile_joint.py 2>&1 | grep -E "host issue|TRAIN step" | sed 's/^/bridge=0: /'
python bench.py --epoch nominal --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); e=d['config']['epoch']; print('epoch eager', e['ms_per_step'], e['passes_ms'], 'joint', {a:(b['ms_per_step'],b.get('passes_ms')) for a,b in e['joint'].items()})"
