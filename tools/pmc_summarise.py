"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into
profiles/<round>/pmc_traffic.json: HBM bytes per launch of every kernel, corrected as MI355X_MICROARCH.md's
HBM section prescribes (counters in KiB; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x -> doubled;
WRITE_SIZE exact).  Usage: python tools/pmc_summarise.py <fetch_dir> <write_dir> <out.json> [workload description]
The file records a digest of the kernel sources it was measured on; bench.py reports `roofline.traffic` from it only
while that digest matches the sources it runs."""
import csv, glob, json, os, re, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def per_kernel(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    acc = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].split("<")[0].replace("void ", "").strip()
            s = acc.setdefault(name, [0.0, 0])
            s[0] += float(r["Counter_Value"]); s[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


def main():
    fetch, write, out = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else "davis_b64 (N=19200, E=57484), fwd+bwd, eager"
    import bench
    fe, wr = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        if k.startswith("__amd") or k.startswith("at::"):
            continue
        f, w = fe.get(k, 0.0), wr.get(k, 0.0)
        kernels[k] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                      "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    doc = {"workload": workload, "kernel_source_digest": bench.kernel_source_digest(),
           "collected": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py --no-graph",
           "correction": "MI355X_MICROARCH.md HBM section: counters in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of "
                         "wide coalesced 16-B/lane reads -> doubled; WRITE_SIZE exact (incl. float atomics)",
           "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:28s} fetch {v['FETCH_SIZE_KiB']:10.1f} KiB  write {v['WRITE_SIZE_KiB']:10.1f} KiB  -> {v['hbm_bytes_per_launch'] / 1e6:8.2f} MB")


if __name__ == "__main__":
    main()
