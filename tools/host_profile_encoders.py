"""Diagnostic: HOST time of an eager encoders training step (both custom ops, forward + backward), davis_b64."""
import cProfile, io, os, pstats, sys, time
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
pd, md = to(ds.to_torch(pb)), to(ds.to_torch(mb))
pp = [p for p in model.protein_gnn.parameters() if p.numel()]
mp = [p for p in model.molecule_gnn.parameters() if p.numel()]
g_res, g_atm = torch.randn(pb.num_nodes, 64, device=dev), torch.randn(mb.num_nodes, 64, device=dev)

def step():                       # what loss.backward() does to the encoders: accumulate mode, then the optimizer's reset
    res = model.protein_gnn(**pd)
    atm = model.molecule_gnn(**md)
    torch.autograd.backward([res, atm], [g_res, g_atm])
    for q in pp + mp:
        q.grad = None

for _ in range(5):
    step()
torch.cuda.synchronize()
for label, fn in (("fwd+bwd", step), ("protein fwd only", lambda: model.protein_gnn(**pd)), ("drug fwd only", lambda: model.molecule_gnn(**md))):
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label:18s}: host issue {1e3 * (t1 - t0) / 50:.3f} ms/step, with final sync {1e3 * (t2 - t0) / 50:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:7000])

from gvp_hip import _lib
if _lib.bridge() is not None and os.environ.get("CGVP_BRIDGE_TIMING"):
    print(_lib.bridge().timing_report())
