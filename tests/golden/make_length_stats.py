#!/usr/bin/env python3
"""Sequence LENGTHS of the Davis / KIBA proteins the reference ships
(/root/reference/data/deepdta_data/{davis,kiba}/proteins.txt: JSON maps id -> sequence).  Only the integer
lengths are written (tests/golden/protein_lengths.json): they size the "real-length" synthetic batches
(BASELINE config 3 = KIBA, 32 pairs per GPU).  Runs only in the build container."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/data/deepdta_data"


def main():
    out = {}
    for name in ("davis", "kiba"):
        seqs = json.load(open(os.path.join(REF, name, "proteins.txt")))
        out[name] = [len(s) for s in seqs.values()]
    json.dump(out, open(os.path.join(HERE, "protein_lengths.json"), "w"))
    for k, v in out.items():
        print(k, len(v), "proteins; min/mean/max", min(v), sum(v) / len(v), max(v))


if __name__ == "__main__":
    main()
