#!/bin/bash
# Re-run the judged bench lines of a round AFTER profiles/<round>/pmc_traffic_*.json exist for the current kernel
# sources, so that every line carries roofline.traffic.  Usage (GPU box): bash tools/bench_lines.sh r03
RND=$1
mkdir -p profiles/$RND
python bench.py > profiles/$RND/bench_davis_b64.json 2> gpurun_out/bl_a.err
python bench.py --workload kiba_b32 > profiles/$RND/bench_kiba_b32.json 2> gpurun_out/bl_b.err
python bench.py --workload long_graph_x64 --steps 30 > profiles/$RND/bench_long_graph_x64.json 2> gpurun_out/bl_c.err
python bench.py --workload bindingdb_b32_44 --dtype bf16 > profiles/$RND/bench_bindingdb_b32_44_bf16.json 2> gpurun_out/bl_d.err
cp profiles/$RND/bench_*.json gpurun_out/
for f in profiles/$RND/bench_davis_b64.json profiles/$RND/bench_kiba_b32.json profiles/$RND/bench_long_graph_x64.json profiles/$RND/bench_bindingdb_b32_44_bf16.json; do
  python -c "
import json,sys
d=json.loads(open('$f').read().strip().split('\n')[-1]); r=d['roofline']
print('$f', d['ms_per_step'], r['frac'], r['avg_us'], r['traffic'])"
done
