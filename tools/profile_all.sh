#!/bin/bash
# The judged artefacts of a round in one GPU-box call: per-workload bench line + rocprofv3 kernel stats + PMC traffic
# + step trace (tools/profile_round.sh), SQ stall counters for the headline workload, and the joint scope.
# Usage: bash tools/profile_all.sh r03
RND=$1
bash tools/profile_round.sh $RND davis_b64 > gpurun_out/prof_${RND}_a.log 2>&1; tail -1 gpurun_out/prof_${RND}_a.log
bash tools/profile_round.sh $RND long_graph_x64 --steps 30 > gpurun_out/prof_${RND}_b.log 2>&1; tail -1 gpurun_out/prof_${RND}_b.log
bash tools/profile_round.sh $RND bindingdb_b32_44 --dtype bf16 > gpurun_out/prof_${RND}_c.log 2>&1; tail -1 gpurun_out/prof_${RND}_c.log
bash tools/profile_round.sh $RND kiba_b32 > gpurun_out/prof_${RND}_d.log 2>&1; tail -1 gpurun_out/prof_${RND}_d.log
bash tools/profile_sq.sh davis_b64 > gpurun_out/prof_${RND}_sq.log 2>&1; tail -3 gpurun_out/prof_${RND}_sq.log
export TMPDIR=/tmp
for extra in "" "--tunable-gemms" "--two-lane-head" "--two-lane-head --tunable-gemms" "--compile --compile-graph"; do
  tag=joint$(echo "$extra" | tr -d ' -' | cut -c1-28)
  python bench.py --scope joint --steps 50 --no-cpu-baseline --epoch off $extra > gpurun_out/bench_davis_b64_$tag.json 2> gpurun_out/bench_$tag.err
  tail -c 400 gpurun_out/bench_davis_b64_$tag.json
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${RND}_joint -o run -- python3 bench.py --scope joint --steps 50 --no-cpu-baseline --epoch off > gpurun_out/prof_${RND}_joint.log 2>&1
find gpurun_out/prof_${RND}_joint -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_davis_b64_joint.csv
find gpurun_out/prof_${RND}_joint -name "*kernel_trace.csv" -delete; rm -rf gpurun_out/prof_${RND}_joint/*/*.db
python bench.py --no-graph --no-cpu-baseline --epoch off > gpurun_out/bench_davis_b64_eager.json 2>/dev/null; tail -c 300 gpurun_out/bench_davis_b64_eager.json
