"""Raw kernel timeline (all queues) of a few consecutive steps in the middle of a `rocprofv3 --kernel-trace
--output-format csv` run of bench.py: every kernel that starts between the k-th and (k+n)-th pass_begin_kernel."""
import csv, glob, os, sys
d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
begins = [i for i, r in enumerate(rows) if "pass_begin_kernel" in r["Kernel_Name"]]
k = len(begins) // 2
lo, hi = begins[k], begins[k + n]
t0 = int(rows[lo]["Start_Timestamp"])
short = lambda s: s.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"q{r['Queue_Id']:>2} {(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} us  {short(r['Kernel_Name'])}")
