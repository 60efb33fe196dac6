"""Diagnostic: who launches the small fill / copy kernels of an eager whole-model training step?  torch.profiler with stacks,
aggregated by (op, innermost repo / torch.autograd frame)."""
import collections, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "caster-dta_amd"), REPO]
import davis_synth as ds
import __graft_entry__ as entry
dev = torch.device("cuda:0")
model, state = entry._load_model(dev)
model.train()
pb, mb = ds.pair_batch(64, 0)
to = lambda d: {k: (tuple(t.to(dev) for t in v) if isinstance(v, tuple) else v.to(dev)) for k, v in d.items()}
p, m = to(ds.to_torch(pb)), to(ds.to_torch(mb))
p["ptr"], m["ptr"] = torch.as_tensor(pb.ptr).to(dev), torch.as_tensor(mb.ptr).to(dev)
params = [q for q in model.parameters() if q.numel()]
target = torch.randn(64, 1, device=dev)
def step():
    pred, _ = model(p, m)
    loss = torch.nn.functional.mse_loss(pred, target)
    return torch.autograd.grad(loss, params, allow_unused=True)
for _ in range(3):
    step()
torch.cuda.synchronize()
names = sys.argv[1:] or ["aten::fill_", "aten::zero_", "aten::zeros", "aten::copy_", "aten::add", "aten::add_", "aten::contiguous"]
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True,
                            experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step()
cnt = collections.Counter()
for e in prof.events():
    if e.name in names:
        frames = [f for f in (e.stack or []) if "caster-dta_amd" in f or "find_fills" in f]
        parent = e.cpu_parent.name if e.cpu_parent is not None else "-"
        gp = e.cpu_parent.cpu_parent.name if (e.cpu_parent is not None and e.cpu_parent.cpu_parent is not None) else "-"
        cnt[(e.name, (frames[0][-70:] if frames else "") + " <- " + parent + " <- " + gp)] += 1
for (n, f), c in cnt.most_common(40):
    print(f"{c:4d}  {n:18s} {f[-110:]}")
