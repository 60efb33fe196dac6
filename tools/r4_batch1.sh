#!/bin/bash
# one GPU-box call: parity of conv backward v2 (gather-ahead) and v3 (two waves per SIMD), then A/B of the three kernels
O=gpurun_out
python -m pytest tests -m gpu -q > $O/r4_gpu5.log 2>&1; rc=$?; echo "pytest(v2) rc=$rc" >> $O/r4_gpu5.log; tail -6 $O/r4_gpu5.log
for i in 1 2 3; do
  CGVP_EXACT_LEAVES=1 python -m pytest tests/test_hip_models.py -m gpu -q -k fused_parameter_mode 2>&1 | tail -1
  python -m pytest tests/test_hip_models.py -m gpu -q -k fused_parameter_mode 2>&1 | tail -1
done > $O/r4_flaky.txt 2>&1; cat $O/r4_flaky.txt
CGVP_CONV_BWD=3 python -m pytest tests/test_hip_backward.py tests/test_hip_random_graphs.py tests/test_hip_configs.py tests/test_bf16_storage.py tests/test_conv_layer_kinds.py -m gpu -q > $O/r4_gpu5_v3.log 2>&1; rc3=$?; echo "pytest(v3) rc=$rc3" >> $O/r4_gpu5_v3.log; tail -4 $O/r4_gpu5_v3.log
bash tools/ab_env.sh CGVP_CONV_BWD "1 2 3" > $O/r4_ab_davis3.txt 2>&1; cat $O/r4_ab_davis3.txt
bash tools/ab_env.sh CGVP_CONV_BWD "1 2 3" --workload long_graph_x64 > $O/r4_ab_long3.txt 2>&1; cat $O/r4_ab_long3.txt
python tools/stamp_conv_bwd.py davis > $O/r4_stamps_davis_v2b.txt 2>&1; cat $O/r4_stamps_davis_v2b.txt
exit $rc
