#!/bin/bash
# Per-kernel times of the captured whole-model training step (rocprofv3 --stats of bench.py --scope joint), per step.
export TMPDIR=/tmp
OUT=gpurun_out/jointk
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 bench.py --scope joint --no-cpu-baseline --epoch off --steps 100 > $OUT/log.txt 2>&1
python - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
steps=max(int(r["Calls"]) for r in rows if "attn_fwd_kernel" in r["Name"])
tot=0
for r in rows[:40]:
    n=r["Name"].replace("void ","").replace("(anonymous namespace)::","").replace("at::native::","")[:70]
    per=float(r["TotalDurationNs"])/steps/1e3; tot+=per
    print("%-72s %6.1f/step %8.1f us avg %8.1f us/step" % (n, int(r["Calls"])/steps, float(r["AverageNs"])/1e3, per))
print("all kernels per step: %.1f us over %d steps" % (sum(float(r["TotalDurationNs"]) for r in rows)/steps/1e3, steps))
PY
cp $f gpurun_out/kernel_stats_joint_latest.csv 2>/dev/null; rm -rf $OUT
