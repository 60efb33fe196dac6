#!/bin/bash
# round-4 judged artefacts, call 2 of 4: long_graph_x64 and the bf16 config
RND=${1:-r04}
bash tools/profile_round.sh $RND long_graph_x64 --steps 30 > gpurun_out/prof_${RND}_b.log 2>&1; tail -1 gpurun_out/prof_${RND}_b.log
bash tools/profile_round.sh $RND bindingdb_b32_44 --dtype bf16 > gpurun_out/prof_${RND}_c.log 2>&1; tail -1 gpurun_out/prof_${RND}_c.log
bash tools/profile_sq.sh davis_b64 > gpurun_out/prof_${RND}_sq.log 2>&1; tail -3 gpurun_out/prof_${RND}_sq.log
bash tools/profile_sq.sh long_graph_x64 --steps 4 > gpurun_out/prof_${RND}_sq_long.log 2>&1; tail -3 gpurun_out/prof_${RND}_sq_long.log
