#!/bin/bash
O=gpurun_out
for cfg in "full none" "full model" "full matmul" "full prot" "prot model" "drug model" "prot_bwd model" "head model" "fwd model" "loss model"; do
  timeout -k 10 120 python -X faulthandler tools/debug_graphed2.py $cfg > $O/r4_dbg2.log 2>&1; echo "[$cfg] rc=$? : $(grep -E '^ok|^start' $O/r4_dbg2.log | tr '\n' ' ')"
done
