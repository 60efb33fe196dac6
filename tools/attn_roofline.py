"""Roofline lines of the three attention kernels at davis_b64 sizes from a rocprofv3 kernel-stats CSV of the joint step
(64 pairs x 300 residues x 40 atoms, 8 heads of 16): useful FLOP and algorithmic HBM bytes per launch over the profiler's
average duration, against the MI355X peaks of MI355X_MICROARCH.md (fp32 matrix 157.3 TFLOP/s, HBM 8 TB/s).

    python tools/attn_roofline.py profiles/r04/kernel_stats_davis_b64_joint.csv > profiles/r04/roofline_attn.json
"""
import csv
import json
import sys

B, R, A, H, D = 64, 300, 40, 8, 16
E = H * D
PEAK_TF, PEAK_HBM = 157.3, 8000.0
pairs = B * R * A                                   # (query, key) pairs per direction, both directions are R x A
rows = B * (R + A)
# forward: S = Q K^T and O = P V per head and direction
f_fwd = 2 * 2 * 2 * pairs * D * H
b_fwd = 4 * E * rows * 4 + 4 * rows * H             # q, k, v read + out written (+ lse) over both directions
# d Q: recompute S, dP = dO V^T, dQ = dS K;  d K / d V: recompute S, dP, dV = P^T dO, dK = dS^T Q
f_dq = 2 * 3 * 2 * pairs * D * H
f_dkv = 2 * 4 * 2 * pairs * D * H
b_dq = 6 * E * rows * 4 + 8 * rows * H              # q, k, v, dO, O read + dQ written (+ lse, delta)
b_dkv = 6 * E * rows * 4 + 8 * rows * H             # q, k, v, dO read + dK, dV written (+ lse, delta)
spec = {"attn_fwd_kernel": (f_fwd, b_fwd), "attn_bwd_dq_kernel": (f_dq, b_dq), "attn_bwd_dkv_kernel": (f_dkv, b_dkv)}
out = []
for r in csv.DictReader(open(sys.argv[1])):
    for k, (fl, by) in spec.items():
        if k in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            out.append({"kernel": k, "workload": "davis_b64 joint step (64 x 300 residues x 40 atoms, 8 heads x 16)",
                        "calls": int(r["Calls"]), "avg_us": round(us, 2),
                        "mfma": {"useful_flop": fl, "achieved": round(fl / us / 1e6, 2), "peak": PEAK_TF, "unit": "TFLOP/s",
                                 "frac": round(fl / us / 1e6 / PEAK_TF, 4)},
                        "hbm": {"algorithmic_bytes": by, "achieved": round(by / us / 1e3, 1), "peak": PEAK_HBM, "unit": "GB/s",
                                "frac": round(by / us / 1e3 / PEAK_HBM, 4)}})
print(json.dumps(out, indent=1))
