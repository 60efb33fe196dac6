#!/bin/bash
# round 4, batch 25: attention kernels (long-wave problem first, v_exp_f32, 32-bit offsets): tests, kernel stats of the joint step
timeout -k 10 300 python -m pytest tests/test_attention.py tests/test_hip_models.py -m gpu -x -q > gpurun_out/gpu_tests_b25.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/gpu_tests_b25.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b25_joint -o run -- python3 bench.py --scope joint --steps 50 --no-cpu-baseline --epoch off > gpurun_out/prof_b25_joint.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_b25_joint/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "attn_" in r["Name"]: print(r["Name"].replace("(anonymous namespace)::","").split("(")[0], r['Calls'], 'avg %.2f us' % (float(r['AverageNs'])/1e3))
PY
rm -rf gpurun_out/prof_b25_joint
python bench.py --scope joint --steps 50 --no-cpu-baseline --epoch off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('joint ms_per_step', d['ms_per_step'])"
