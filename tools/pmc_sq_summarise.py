"""Summarise rocprofv3 --pmc SQ_* passes (separate runs of bench.py --no-graph) into profiles/<round>/pmc_sq_stalls.json:
per kernel, the per-launch average of every counter found plus a few derived fractions.
Usage: python tools/pmc_sq_summarise.py <out.json> <pass_dir> [<pass_dir> ...]"""
import csv, glob, json, os, re, sys


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].split("<")[0].replace("void ", "").strip()
                if name.startswith("__amd") or name.startswith("at::"):
                    continue
                s = acc.setdefault(name, {}).setdefault(r["Counter_Name"], [0.0, 0])
                s[0] += float(r["Counter_Value"]); s[1] += 1
    kernels = {}
    for k, cs in sorted(acc.items()):
        v = {c: round(t / n) for c, (t, n) in cs.items()}
        wc = v.get("SQ_WAVE_CYCLES")
        if wc:
            for src, dst in (("SQ_WAIT_INST_ANY", "_wait_inst_any_frac"), ("SQ_WAIT_ANY", "_wait_any_frac"),
                             ("SQ_ACTIVE_INST_ANY", "_active_frac")):
                if src in v:
                    v[dst] = round(v[src] / wc, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CYCLES" in v and v["SQ_BUSY_CYCLES"]:
            v["_mfma_busy_over_sq_busy"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_BUSY_CYCLES"], 3)
        kernels[k] = v
    tag = os.path.basename(out).replace("pmc_sq_stalls_", "").replace(".json", "")      # workload (and dtype) of this file
    json.dump({"note": f"rocprofv3 --pmc SQ_* (separate passes), {tag} fwd+bwd eager, per-launch averages; raw counter "
                       "units as rocprofv3 reports them on gfx950 (WAVE/WAIT/ACTIVE in quad-cycles summed over waves)",
               "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(k, {a: b for a, b in v.items() if a.startswith("_")})


if __name__ == "__main__":
    main()
