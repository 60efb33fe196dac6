#!/bin/bash
# round-4 judged artefacts, call 3 of 4: the joint scope (captured step, its variants, kernel stats) and the eager headline
RND=${1:-r04}
export TMPDIR=/tmp
for extra in "" "--two-lane-head" "--tunable-gemms" "--compile --compile-graph"; do
  tag=joint$(echo "$extra" | tr -d ' -' | cut -c1-28)
  python bench.py --scope joint --steps 50 --no-cpu-baseline --epoch off $extra > gpurun_out/bench_davis_b64_$tag.json 2> gpurun_out/bench_$tag.err
  tail -c 300 gpurun_out/bench_davis_b64_$tag.json; echo
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${RND}_joint -o run -- python3 bench.py --scope joint --steps 50 --no-cpu-baseline --epoch off > gpurun_out/prof_${RND}_joint.log 2>&1
find gpurun_out/prof_${RND}_joint -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_davis_b64_joint.csv
find gpurun_out/prof_${RND}_joint -name "*kernel_trace.csv" -delete; rm -rf gpurun_out/prof_${RND}_joint/*/*.db
python bench.py --no-graph --no-cpu-baseline --epoch off > gpurun_out/bench_davis_b64_eager.json 2>/dev/null; tail -c 300 gpurun_out/bench_davis_b64_eager.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${RND}_cpd -o run -- python3 tools/cpd_fwdbwd.py 20 > gpurun_out/cpd_fwdbwd.txt 2>&1
find gpurun_out/prof_${RND}_cpd -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_cpdmodel.csv
find gpurun_out/prof_${RND}_cpd -name "*kernel_trace.csv" -delete; rm -rf gpurun_out/prof_${RND}_cpd/*/*.db
tail -2 gpurun_out/cpd_fwdbwd.txt
