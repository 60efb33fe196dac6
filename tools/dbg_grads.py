import sys, os
sys.path[:0] = ["/root/repo/tests", "/root/repo/caster-dta_amd", "/root/repo"]
import numpy as np, torch
import test_hip_backward as tb
g = dict(np.load("/root/repo/tests/golden/lba_small.npz"))
st = dict(np.load("/root/repo/tests/golden/pretrained_state.npz"))
T = torch.from_numpy
pp = {k[len("protein_gnn.gnn_model."):]: T(v) for k, v in st.items() if k.startswith("protein_gnn.gnn_model.")}
model = tb._encoder(pp).eval()
DEV = tb.DEV
xs, xv = T(g["x_s"]).to(DEV).requires_grad_(), T(g["x_v"]).to(DEV).requires_grad_()
out = model((xs, xv), T(g["edge_index"]).to(DEV), T(g["ntypes"]).to(DEV), T(g["etypes"]).to(DEV),
            eattr=(T(g["e_s"]).to(DEV), T(g["e_v"]).to(DEV)), batch=T(g["batch"]).to(DEV))
(out * T(g["r"]).to(DEV)).sum().backward()
for name, p in model.gnn_model.named_parameters():
    if not p.numel(): continue
    ref = T(g["g_" + name])
    err = float((p.grad.cpu() - ref).abs().max())
    print(f"{name:45s} err {err:10.3e} refmax {float(ref.abs().max()):10.3e}")
print("x_s", float((xs.grad.cpu() - T(g["gin_x_s"])).abs().max()))
