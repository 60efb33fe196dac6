"""Diagnostic: device time of the varlen cross-attention forward at davis_b64 sizes, per direction and for
variations of the problem (HIP events around repeated launches of caster_gvp::cross_attention)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "caster-dta_amd"), REPO]
from gvp_hip import attention_ops  # noqa
dev = torch.device("cuda:0")


def run(B, R, A, heads=8, reps=30, label=""):
    E = 16 * heads
    g = torch.Generator(device=dev).manual_seed(0)
    f = lambda n: torch.randn(n, E, device=dev, generator=g)
    q_r, k_r, v_r = f(B * R), f(B * R), f(B * R)
    q_a, k_a, v_a = f(B * A), f(B * A), f(B * A)
    rptr = torch.arange(0, B * R + 1, R, device=dev)
    aptr = torch.arange(0, B * A + 1, A, device=dev)
    for _ in range(5):
        torch.ops.caster_gvp.cross_attention(q_r, k_a, v_a, q_a, k_r, v_r, rptr, aptr, heads)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        torch.ops.caster_gvp.cross_attention(q_r, k_a, v_a, q_a, k_r, v_r, rptr, aptr, heads)
    b.record()
    b.synchronize()
    print(f"{label:40s} B={B:4d} R={R:5d} A={A:4d}: {a.elapsed_time(b) / reps * 1e3:8.1f} us per forward (incl. host gaps)")


run(64, 300, 40, label="davis_b64")
run(64, 300, 1, label="one atom per drug (dir 2 nearly empty)")
run(64, 16, 40, label="16 residues (dir 1 nearly empty)")
run(64, 300, 16, label="16 atoms: one key tile / one query tile")
run(1, 300, 40, label="one pair")
run(64, 1200, 40, label="4x longer proteins")
run(256, 300, 40, label="4x more pairs")
