"""Diagnostic: where the HOST time of an eager JointGNN training step goes (cProfile over 30 steps, davis_b64)."""
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
pd["ptr"], md["ptr"] = torch.as_tensor(pb.ptr).to(dev), torch.as_tensor(mb.ptr).to(dev)
params = [p for p in model.parameters() if p.numel()]
target = torch.randn(64, 1, device=dev)

def step():
    pred, _ = model(pd, md)
    loss = torch.nn.functional.mse_loss(pred, target)
    return torch.autograd.grad(loss, params)

for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    step()
t1 = time.perf_counter()            # host issue time (no sync inside)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"30 steps: host issue {1e3 * (t1 - t0) / 30:.2f} ms/step, with final sync {1e3 * (t2 - t0) / 30:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])

# ---- the TRAINING step on the host: forward, mse loss, backward(), zero_grad -- what train_model.py:548-587 issues
model.train()
target = torch.randn(64, 1, device=dev)


def train_step():
    pred, _ = model(pd, md)
    torch.nn.functional.mse_loss(pred, target).backward()
    model.zero_grad(set_to_none=True)


with torch.enable_grad():
    for _ in range(5):
        train_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        train_step()
    ti = (time.perf_counter() - t0) / 20 * 1e3
    torch.cuda.synchronize()
    print(f"TRAIN step: host issue {ti:.3f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        train_step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
    print(s.getvalue()[:7000])
