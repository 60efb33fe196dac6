#!/bin/bash
# One GPU-box call that produces the judged artefacts of a round for one workload:
#   bench JSON line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE PMC passes (separate runs).
# Usage (on the GPU box, from the repo root):  bash tools/profile_round.sh <round> <workload> [extra bench args]
set -e
RND=$1; WL=$2; shift 2
TAG=$WL
case " $* " in *" bf16 "*) TAG=${WL}_bf16;; esac
OUT=gpurun_out/prof_${RND}_${TAG}
mkdir -p "$OUT" profiles/$RND
export TMPDIR=/tmp
python bench.py --workload $WL "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 bench.py --workload $WL --no-cpu-baseline --epoch off --steps 100 "$@" > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 bench.py --workload $WL --no-cpu-baseline --epoch off --no-graph --steps 10 --warmup 2 "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 bench.py --workload $WL --no-cpu-baseline --epoch off --no-graph --steps 10 --warmup 2 "$@" > $OUT/write.log 2>&1
python tools/pmc_summarise.py $OUT/fetch $OUT/write $OUT/pmc_traffic_${TAG}.json "$TAG fwd+bwd train mode, eager" > $OUT/pmc.txt
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_${TAG}.csv
python tools/step_trace.py $OUT/stats > $OUT/step_trace_${TAG}.txt 2>/dev/null || true
cp $OUT/bench.json $OUT/bench_${TAG}.json
rm -rf $OUT/stats/*/*.db $OUT/fetch $OUT/write 2>/dev/null || true
find $OUT/stats -name "*kernel_trace.csv" -delete 2>/dev/null || true
ls $OUT
