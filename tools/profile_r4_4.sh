#!/bin/bash
# round-4 judged artefacts, call 4 of 4: epoch legs (encoders + joint, nominal and real lengths), host profile, conv backward shapes
RND=${1:-r04}
export TMPDIR=/tmp
python bench.py --epoch nominal --no-cpu-baseline > gpurun_out/bench_davis_b64_epoch_nominal.json 2> gpurun_out/bench_epoch_nominal.err; tail -c 600 gpurun_out/bench_davis_b64_epoch_nominal.json; echo
python bench.py --epoch real --no-cpu-baseline > gpurun_out/bench_davis_b64_epoch_real.json 2> gpurun_out/bench_epoch_real.err; tail -c 600 gpurun_out/bench_davis_b64_epoch_real.json; echo
python tools/host_profile_encoders.py > gpurun_out/host_profile_encoders.txt 2>&1; head -12 gpurun_out/host_profile_encoders.txt
for V in 3 2 1; do
  for WL in "davis_b64 --steps 100" "long_graph_x64 --steps 20"; do
    D=gpurun_out/prof_${RND}_cb$V; rm -rf $D
    CGVP_CONV_BWD=$V rocprofv3 --kernel-trace --stats --output-format csv -d $D -o run -- python3 bench.py --workload $WL --no-cpu-baseline --epoch off > $D.log 2>&1
    f=$(find $D -name "*kernel_stats.csv" | head -1)
    python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'conv_bwd' in r['Name']:
        print('CGVP_CONV_BWD=$V $WL:', r['Name'].split('::')[-1].split('(')[0], 'calls', r['Calls'], 'avg %.2f us' % (float(r['AverageNs']) / 1e3))
"
    rm -rf $D
  done
done > gpurun_out/conv_bwd_shapes.txt 2>&1
cat gpurun_out/conv_bwd_shapes.txt
