#!/bin/bash
# SQ occupancy / stall / MFMA-busy counters of one workload, one counter group per rocprofv3 pass (separate runs, only
# --pmc): -> gpurun_out/prof_sq_<tag>/pmc_sq_stalls_<tag>.json.  Usage: bash tools/profile_sq.sh <workload> [bench args]
set -e
WL=$1; shift
TAG=$WL
case " $* " in *" bf16 "*) TAG=${WL}_bf16;; esac
OUT=gpurun_out/prof_sq_${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i + 1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o run -- python3 bench.py --workload $WL --no-cpu-baseline --epoch off --no-graph --steps 6 --warmup 2 "$@" > $OUT/p$i.log 2>&1 || echo "pass $i ($grp) failed"
done
python tools/pmc_sq_summarise.py $OUT/pmc_sq_stalls_${TAG}.json $OUT/p* | grep -E "conv_bwd|conv_quad|node_bwd|edge_bwd"
rm -rf $OUT/p[0-9]
