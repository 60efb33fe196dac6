"""Debug: GraphedTrainStep on small batches, one configuration per process (a crash in one does not hide the others)."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "caster-dta_amd"), REPO]
import numpy as np, torch
import davis_synth as ds
from gvp_hip.graphed import GraphedTrainStep
from models.joint_gnn import JointGNN
mode, pairs, length = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
GOLDEN = os.path.join(REPO, "tests", "golden")
kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
state = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "pretrained_state.npz")).items()}
model = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"], **kw["joint_gnn_kwargs"])
model.load_state_dict(state, strict=True)
model.to("cuda:0").train(mode == "train")
if len(sys.argv) > 4 and sys.argv[4] == "nofast":
    from gvp_hip import head_ops
    head_ops.FAST_LINEAR_MIN_ROWS = 1 << 60 if hasattr(head_ops, "FAST_LINEAR_MIN_ROWS") else None
p, m = ds.pair_batch(pairs, 1, lengths=[length + 7 * i for i in range(pairs)])
to = lambda d: {k: (tuple(t.to("cuda:0") for t in v) if isinstance(v, tuple) else v.to("cuda:0")) for k, v in d.items()}
pd, md = to(ds.to_torch(p)), to(ds.to_torch(m))
pd["ptr"], md["ptr"] = torch.as_tensor(p.ptr).to("cuda:0"), torch.as_tensor(m.ptr).to("cuda:0")
target = torch.randn(pairs, 1, device="cuda:0")
runner = GraphedTrainStep(model, torch.nn.functional.mse_loss)
pre = sys.argv[4] if len(sys.argv) > 4 else "none"
params = [q for q in model.parameters() if q.numel()]
if pre in ("fwd", "grad", "grad_sync", "backward"):
    pred, _ = model(pd, md)
    loss = torch.nn.functional.mse_loss(pred, target)
    if pre in ("grad", "grad_sync"):
        ref = torch.autograd.grad(loss, params)
    if pre == "backward":
        loss.backward()
    if pre == "grad_sync":
        torch.cuda.synchronize()
    model.zero_grad(set_to_none=True)
print("start", mode, pairs, length, pre, flush=True)
loss, pred = runner.run(pd, md, target)
torch.cuda.synchronize()
print("ok", float(loss), flush=True)
