#!/usr/bin/env python3
"""Instruction mix of one kernel in a gfx950 .s file (hipcc -save-temps): counts by class for the whole kernel and for
its largest loop body (the tile loop).  Usage: tools/isa_mix.py <file.s> <kernel-name-substring>"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_read:" + op
    if op.startswith("ds_write") or op.startswith("ds_store"): return "ds_write:" + op
    if op.startswith("ds_"): return "ds_other"
    if op.startswith("global_atomic") or op.startswith("buffer_atomic"): return "atomic"
    if op.startswith("global_load") or op.startswith("buffer_load"): return "vmem_load"
    if op.startswith("global_store") or op.startswith("buffer_store"): return "vmem_store"
    if op.startswith("scratch_"): return "scratch"
    if op == "s_waitcnt": return "s_waitcnt"
    if op == "s_nop": return "s_nop"
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith("v_cndmask"): return "v_cndmask"
    if op.startswith("v_mov"): return "v_mov"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_0-9$.]*%s[A-Za-z_0-9$.]*:" % re.escape(name), l) and not l.startswith("."):
            start = i
            break
    if start is None:
        cands = [l for l in lines if name in l and l.endswith(":")]
        raise SystemExit("kernel not found; candidates: %s" % cands[:5])
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    # basic blocks
    blocks, cur, label = [], [], "entry"
    for l in body[1:]:
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            blocks.append((label, cur))
            cur, label = [], m.group(1)
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur.append(t.split()[0])
    blocks.append((label, cur))
    total = Counter()
    for _, ops in blocks:
        for op in ops:
            total[classify(op)] += 1
    big = max(blocks, key=lambda b: len(b[1]))

    def show(title, cnt):
        n = sum(cnt.values())
        print("%s: %d instructions" % (title, n))
        agg = Counter()
        for k, v in cnt.items():
            agg[k.split(":")[0]] += v
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
            print("   %-12s %5d" % (k, v))
        for k, v in sorted(cnt.items()):
            if ":" in k:
                print("      %-28s %5d" % (k.split(":")[1], v))
    show("whole kernel", total)
    c = Counter(classify(op) for op in big[1])
    show("largest block %s" % big[0], c)


main()
