"""Forward + backward of the default CPD-style stack (protein_gnn.py:520-606 of the reference: encoder GVPConvLayers,
autoregressive decoder with edge width 32 + 20 = 52) on a davis-like batch, for a rocprofv3 kernel-stats line:

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o run -- python3 tools/cpd_fwdbwd.py [steps]

Prints the wall time per step (events around the timed loop) and which conv layers ran on the tile kernels."""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "caster-dta_amd"))
import davis_synth as ds                                            # noqa: E402
from gvp_hip import conv_layer_ops as clo                           # noqa: E402
from models.protein_gnn import SelectableProteinModelWrapper        # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda:0"
kw = dict(in_channels=(17, 3), edge_dim=(32, 1), num_ntypes=20, num_etypes=1, num_convs=2, hidden_channels=(16, 4),
          dropout_rate=0.1, base_conv="cpdmodel", ntype_emb_dim=None, etype_emb_dim=None, out_channels=8,
          edge_hidden_channels=(32, 1))
torch.manual_seed(0)
model = SelectableProteinModelWrapper(**kw).to(dev).eval()
gb = ds.protein_batch(16, 7)                                        # 16 x 300 residues, ~14k edges
d = ds.to_torch(gb)
to = lambda t: t.to(dev)
x = (to(d["x"][0]), to(d["x"][1]))
args = dict(x=x, edge_index=to(d["edge_index"]), ntypes=to(d["ntypes"]), etypes=to(d["etypes"]),
            eattr=(to(d["eattr"][0]), to(d["eattr"][1])), batch=to(d["batch"]))
params = [p for p in model.parameters() if p.numel()]
calls = {"kernels": 0}
orig = clo.conv_kind


def counting(conv):
    k = orig(conv)
    calls["kernels"] += k is not None
    return k


clo.conv_kind = counting


def step():
    out = model(**args)
    torch.autograd.grad(out.sum(), params, allow_unused=True)


for _ in range(3):
    step()
calls["kernels"] = 0
step()
per_step = calls["kernels"]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / steps
print(f"cpdmodel (default widths) fwd+bwd, {gb.num_nodes} residues / {gb.num_edges} edges, eager: {ms:.3f} ms per step; "
      f"conv-layer dispatches that chose the tile kernels per step: {per_step}")
