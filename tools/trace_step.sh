#!/bin/bash
# Kernel timeline (all queues) of two consecutive steps of the captured encoders step: rocprofv3 kernel trace of a short
# bench run, then tools/trace_window.py.  Usage (GPU box): bash tools/trace_step.sh <tag> [bench args]
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 bench.py --no-cpu-baseline --epoch off --steps 40 --warmup 10 "$@" > $OUT/bench.log 2>&1
python tools/trace_window.py $OUT 2 > gpurun_out/trace_$TAG.txt
find $OUT -name "*kernel_trace.csv" -delete; rm -rf $OUT
tail -1 gpurun_out/trace_$TAG.txt
