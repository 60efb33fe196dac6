#!/bin/bash
# round 4, batch 30: two-lane head by default while a step is being captured
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_b30.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/gpu_tests_b30.log
python bench.py --scope joint --steps 50 --no-cpu-baseline --epoch off 2>gpurun_out/b30_joint.err | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('joint captured (default flags) ms_per_step', d['ms_per_step'])"
python bench.py --epoch nominal --no-cpu-baseline 2>gpurun_out/b30_epoch.err > gpurun_out/bench_davis_b64_epoch_nominal.json; python -c "
import json
d=json.loads(open('gpurun_out/bench_davis_b64_epoch_nominal.json').read().strip().split('\n')[-1]); e=d['config']['epoch']; print('joint epoch', {a:(b['ms_per_step'],b.get('passes_ms')) for a,b in e['joint'].items()})"
