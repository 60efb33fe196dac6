#!/bin/bash
# round 4, batch 19: one-launch CSR build for small batches (inside pass_begin for the protein pass) -- tests, A/B, trace
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_b19.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/gpu_tests_b19.log
for rep in 1 2; do
for O in 1 0; do
for args in "" "--only protein" "--only drug"; do
  CGVP_CSR_OWNER=$O python bench.py --no-cpu-baseline --epoch off --steps 300 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('csr_owner=$O %-20s ms_per_step %.4f' % ('$args' or 'default', d['ms_per_step']))"
done
done
done
bash tools/trace_step.sh b19
