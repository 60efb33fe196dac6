#!/bin/bash
O=gpurun_out
python -X faulthandler -m pytest tests -m gpu -q --deselect tests/test_hip_models.py::test_graphed_train_step_equals_eager > $O/r4_gpu7.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/r4_gpu7.log; grep -E "^FAILED|passed|failed|^E  " $O/r4_gpu7.log | cut -c1-300 | head -40
for v in compare searchsorted; do
  CGVP_GRAPHED_BATCH=$v timeout -k 10 300 python -X faulthandler -m pytest tests/test_hip_models.py -m gpu -q -k graphed_train_step > $O/r4_graphed_$v.log 2>&1; echo "graphed($v) rc=$?"; tail -3 $O/r4_graphed_$v.log | cut -c1-300
done
exit $rc
