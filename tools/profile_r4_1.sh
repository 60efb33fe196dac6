#!/bin/bash
# round-4 judged artefacts, call 1 of 4: headline workload + kiba_b32 (bench line, rocprofv3 kernel stats, PMC traffic, step trace)
RND=${1:-r04}
bash tools/profile_round.sh $RND davis_b64 > gpurun_out/prof_${RND}_a.log 2>&1; tail -1 gpurun_out/prof_${RND}_a.log
bash tools/profile_round.sh $RND kiba_b32 > gpurun_out/prof_${RND}_d.log 2>&1; tail -1 gpurun_out/prof_${RND}_d.log
bash tools/chain_split.sh > gpurun_out/chain_split_davis_b64.txt 2>&1; cat gpurun_out/chain_split_davis_b64.txt
