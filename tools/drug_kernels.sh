#!/bin/bash
# Per-kernel times of the drug chain alone (rocprofv3 --stats of bench.py --only drug).  Usage: bash tools/drug_kernels.sh <tag>
export TMPDIR=/tmp
OUT=gpurun_out/drugk_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 bench.py --only drug --no-cpu-baseline --epoch off --steps 200 > $OUT/log.txt 2>&1
python - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"].replace("void ","").replace("(anonymous namespace)::","")[:60]
    print("%-62s %6s %9.1f" % (n, r["Calls"], float(r["AverageNs"])/1e3))
PY
rm -rf $OUT
