"""Timeline of ONE step from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py.

    python tools/step_trace.py <dir with *_kernel_trace.csv> [nth-from-last step; default: the shortest step,
                                                               that runs both encoders, i.e. one HIP-graph
                                                               replay of the timed region]

Prints every kernel of the chosen step with start / end relative to the step's first kernel, its queue, and the gap to
the previous kernel on the same queue.  A step is delimited by `reduce_segments_kernel` (the last launch of the protein
backward).
"""
import csv, glob, os, sys

d = sys.argv[1]
nth = int(sys.argv[2]) if len(sys.argv) > 2 else None
paths = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for p in paths:
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "reduce_segments_kernel" in r["Kernel_Name"]]
if nth is None:
    # candidates: steps that run BOTH encoders (bench.py's roofline leg launches protein-only passes: not a step) --
    # of those, the shortest is a HIP-graph replay of the timed region
    span = lambda k: int(rows[ends[k]]["End_Timestamp"]) - int(rows[ends[k - 1]]["End_Timestamp"])
    both = [k for k in range(2, len(ends) - 1)
            if any("gine_quad_bwd" in r["Kernel_Name"] for r in rows[ends[k - 1] + 1:ends[k] + 1])]
    nth = len(ends) - min(both or range(2, len(ends) - 1), key=span)
hi, lo = ends[-nth], ends[-nth - 1]
t_prev_end = int(rows[lo]["End_Timestamp"])
step = [r for r in rows[lo + 1:] if int(r["Start_Timestamp"]) <= int(rows[hi]["End_Timestamp"])]
t0 = int(step[0]["Start_Timestamp"])
last_end = {}
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
print(f"step = {(int(rows[hi]['End_Timestamp']) - t0) / 1e3:.1f} us first-start..last-end; "
      f"{(int(rows[hi]['End_Timestamp']) - t_prev_end) / 1e3:.1f} us end-to-end of consecutive steps")
for r in step:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    last_end[q] = e
    print(f"q{q:>2} {(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  "
          f"grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>4}  {short(r['Kernel_Name'])}")
