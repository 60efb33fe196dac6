"""Debug: which part of a captured JointGNN step crashes hipStreamEndCapture after an eager model call?  One stage per process."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "caster-dta_amd"), REPO]
import numpy as np, torch
import davis_synth as ds
from models.joint_gnn import JointGNN
stage, pre = sys.argv[1], sys.argv[2]
GOLDEN = os.path.join(REPO, "tests", "golden")
kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
state = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "pretrained_state.npz")).items()}
model = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"], **kw["joint_gnn_kwargs"])
model.load_state_dict(state, strict=True)
model.to("cuda:0").train()
model.attention_weights = "never"
pairs = 6
p, m = ds.pair_batch(pairs, 1, lengths=[100 + 7 * i for i in range(pairs)])
to = lambda d: {k: (tuple(t.to("cuda:0") for t in v) if isinstance(v, tuple) else v.to("cuda:0")) for k, v in d.items()}
pd, md = to(ds.to_torch(p)), to(ds.to_torch(m))
pd["ptr"], md["ptr"] = torch.as_tensor(p.ptr).to("cuda:0"), torch.as_tensor(m.ptr).to("cuda:0")
target = torch.randn(pairs, 1, device="cuda:0")
params = [q for q in model.parameters() if q.numel()]
enc = lambda d: {k: v for k, v in d.items() if k != "ptr"}

def step():
    if stage == "prot":
        return model.protein_gnn(**enc(pd))
    if stage == "drug":
        return model.molecule_gnn(**enc(md))
    if stage == "prot_bwd":
        r = model.protein_gnn(**enc(pd))
        return torch.autograd.grad(r.sum(), [q for q in model.protein_gnn.parameters() if q.numel()])
    if stage == "head":
        with torch.no_grad():
            r, a = model.protein_gnn(**enc(pd)), model.molecule_gnn(**enc(md))
        return model.head(r, a, pd, md)[0]
    if stage == "fwd":
        return model(pd, md)[0]
    if stage == "loss":
        return torch.nn.functional.mse_loss(model(pd, md)[0], target)
    if stage == "full":
        return torch.autograd.grad(torch.nn.functional.mse_loss(model(pd, md)[0], target), params)
    raise SystemExit("stage?")

if pre == "model":
    model(pd, md)
elif pre == "prot":
    model.protein_gnn(**enc(pd))
elif pre == "matmul":
    (torch.randn(64, 64, device="cuda:0") @ torch.randn(64, 64, device="cuda:0")).sum().item()
elif pre == "sync":
    torch.cuda.synchronize()
print("start", stage, pre, flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s), torch.enable_grad():
    for _ in range(2):
        step()
    torch.cuda.current_stream().synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = step()
    g.replay()
torch.cuda.synchronize()
print("ok", flush=True)
