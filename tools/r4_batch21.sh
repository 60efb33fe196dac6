#!/bin/bash
# round 4, batch 21: edge stage with register-resident weight gradients: merged tail launch vs two launches
run() { python bench.py --no-cpu-baseline --epoch off "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('split_tail=$T %-50s ms_per_step %.4f' % ('$*', d['ms_per_step']))"; }
for rep in 1 2; do
for T in 0 1; do
  export CGVP_SPLIT_TAIL_BWD=$T
  run --steps 300
  run --steps 30 --workload long_graph_x64
done
done
unset CGVP_SPLIT_TAIL_BWD; T=default
run --steps 300 --workload kiba_b32
run --steps 300 --workload bindingdb_b32_44 --dtype bf16
timeout -k 10 600 python -m pytest tests -m gpu -x -q tests/test_hip_backward.py tests/test_hip_parity.py tests/test_hip_configs.py tests/test_bf16_storage.py tests/test_hip_random_graphs.py > gpurun_out/gpu_tests_b21.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/gpu_tests_b21.log
