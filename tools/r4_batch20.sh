#!/bin/bash
# round 4, batch 20: HEAD after the CSR-owner experiment was removed + d(edge embedding) in the storage type
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_b20.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/gpu_tests_b20.log
run() { python bench.py --no-cpu-baseline --epoch off "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('%-50s ms_per_step %.4f  dominant %s us' % ('$*', d['ms_per_step'], r.get('avg_us')))"; }
for rep in 1 2; do
  run --steps 300
  run --steps 300 --workload bindingdb_b32_44 --dtype bf16
done
run --steps 300 --workload kiba_b32
run --steps 30 --workload long_graph_x64
