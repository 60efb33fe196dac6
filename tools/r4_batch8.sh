#!/bin/bash
O=gpurun_out
python -X faulthandler -m pytest tests -m gpu -q > $O/r4_gpu9.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/r4_gpu9.log; grep -E "^FAILED|passed|failed|^E  |Fatal" $O/r4_gpu9.log | cut -c1-300 | head -20
bash tools/profile_all.sh r04 2>&1 | cut -c1-400
bash tools/profile_sq.sh long_graph_x64 --steps 3 2>&1 | tail -5
exit $rc
