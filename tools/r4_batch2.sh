#!/bin/bash
O=gpurun_out
python -m pytest tests -m gpu -q > $O/r4_gpu6.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/r4_gpu6.log; tail -6 $O/r4_gpu6.log
bash tools/ab_env.sh CGVP_CONV_BWD "1 2 3" --config 5 > $O/r4_ab_c5.txt 2>&1; cat $O/r4_ab_c5.txt
bash tools/ab_env.sh CGVP_CONV_BWD "1 3" --workload kiba_b32 > $O/r4_ab_kiba.txt 2>&1; cat $O/r4_ab_kiba.txt
python tools/stamp_conv_bwd.py davis > $O/r4_stamps_davis_v3.txt 2>&1; cat $O/r4_stamps_davis_v3.txt
python tools/stamp_conv.py > $O/r4_stamps_fwd.txt 2>&1; tail -25 $O/r4_stamps_fwd.txt
python tools/stamp_node_bwd.py > $O/r4_stamps_node.txt 2>&1; tail -25 $O/r4_stamps_node.txt
python bench.py > $O/r4_bench2.json 2> $O/r4_bench2.err; tail -c 3000 $O/r4_bench2.json
python bench.py --workload long_graph_x64 --no-cpu-baseline > $O/r4_bench2_long.json 2> $O/r4_bench2_long.err; tail -c 1500 $O/r4_bench2_long.json
exit $rc
