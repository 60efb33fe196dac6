"""Diagnostic: how far are the bf16 kernels' outputs / weight gradients from (a) the oracle that emulates the same
roundings (store_dtype + emulate_gemm_dtype: the test's checker), (b) the oracle that only rounds the stored tensors,
(c) the unrounded fp32 oracle -- per parameter, L2-relative.  Run from the repo root on a GPU box."""
import os, sys, json, torch
sys.path[:0] = [os.path.join(os.getcwd(), "caster-dta_amd"), os.getcwd(), os.path.join(os.getcwd(), "tests")]
import davis_synth as ds
from oracle import gvp_oracle as O
import test_bf16_storage as T
from conftest import rel_err
import numpy as np
_z = np.load("tests/golden/pretrained_state.npz")
state = {k[len("protein_gnn.gnn_model."):]: torch.from_numpy(_z[k]) for k in _z.files if k.startswith("protein_gnn.gnn_model.")}
DEV = "cuda:0"
gb = ds.protein_batch(8, 3)
model = T._encoder(state, 2, seed=11)
d = ds.to_torch(gb)
dd, dc = T._bf(d, DEV), T._bf(d)
def run_kernel(inp):
    model.zero_grad()
    xs, xv = inp["x"][0].clone().requires_grad_(), inp["x"][1].clone().requires_grad_()
    out = model((xs, xv), inp["edge_index"], inp["ntypes"], inp["etypes"], eattr=inp["eattr"])
    r = torch.randn(out.shape, generator=torch.Generator().manual_seed(3))
    (out.float() * r.to(DEV)).sum().backward()
    return out.float().detach().cpu(), {n: p.grad.detach().cpu().clone() for n, p in model.gnn_model.named_parameters() if p.numel()}
def run_oracle(emu, store):
    P = {k: v.detach().cpu().clone().requires_grad_(v.numel() > 0) for k, v in model.gnn_model.state_dict().items()}
    import contextlib
    cm = O.emulate_gemm_dtype(torch.bfloat16) if emu else contextlib.nullcontext()
    with cm:
        ref = O.protein_lba_forward(P, (dc["x"][0].clone(), dc["x"][1].clone()), dc["edge_index"], dc["ntypes"], dc["etypes"], dc["eattr"], num_convs=2, store_dtype=store)
        r = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3))
        (ref * r).sum().backward()
    return ref.detach(), {k: v.grad for k, v in P.items() if v.grad is not None}
o16, g16 = run_kernel(dd)
f32 = {k: (tuple(t.float() for t in v) if isinstance(v, tuple) else v) for k, v in dd.items()}
o32, g32 = run_kernel(f32)
refs = {"emu+store": run_oracle(True, torch.bfloat16), "store only": run_oracle(False, torch.bfloat16), "fp32 oracle": run_oracle(False, None)}
l2 = lambda a, b: float((a - b).norm() / b.norm())
print("forward: bf16 kernel vs", {k: round(l2(o16, v[0]), 5) for k, v in refs.items()}, " fp32 kernel vs fp32 oracle", round(l2(o32, refs["fp32 oracle"][0]), 7))
print("oracle emu+store vs fp32 oracle (fwd)", round(l2(refs["emu+store"][0], refs["fp32 oracle"][0]), 5))
names = list(g16)
for n in names:
    print(f"{n:55s} " + "  ".join(f"{k}: {l2(g16[n], v[1][n]):.4f}" for k, v in refs.items()) + f"   | emu vs fp32-oracle {l2(refs['emu+store'][1][n], refs['fp32 oracle'][1][n]):.4f}")
