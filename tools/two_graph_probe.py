"""Diagnostic: the captured encoders step as ONE two-branch HIP graph (what bench.py times) vs TWO graphs, one per encoder,
replayed on two streams with an event join per step -- is the ~15 us the second chain adds the graph's own fork / join?"""
import os, sys, time
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
A, B = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def protein():
    return torch.autograd.grad([model.protein_gnn(**pd)], pp, [g_res])


def drug():
    return torch.autograd.grad([model.molecule_gnn(**md)], mp, [g_atm])


def both_one_stream_pair():
    B.wait_stream(A)
    gp = protein()
    with torch.cuda.stream(B):
        gd = drug()
    A.wait_stream(B)
    return gp, gd


with torch.cuda.stream(A):
    for _ in range(5):
        both_one_stream_pair()
torch.cuda.synchronize()
# (1) one graph, two branches
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=A):
    keep1 = both_one_stream_pair()
# (2) two graphs
gP, gD = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.stream(B):
    for _ in range(3):
        drug()
torch.cuda.synchronize()
with torch.cuda.graph(gP, stream=A):
    keepP = protein()
with torch.cuda.graph(gD, stream=B):
    keepD = drug()
torch.cuda.synchronize()


def timed(fn, K=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


def one():
    with torch.cuda.stream(A):
        g1.replay()


def two_joined():
    B.wait_stream(A)
    with torch.cuda.stream(A):
        gP.replay()
    with torch.cuda.stream(B):
        gD.replay()
    A.wait_stream(B)


def two_free():
    with torch.cuda.stream(A):
        gP.replay()
    with torch.cuda.stream(B):
        gD.replay()


def protein_only():
    with torch.cuda.stream(A):
        gP.replay()


for rep in range(2):
    print(f"one two-branch graph      : {timed(one):.4f} ms/step")
    print(f"two graphs, join per step : {timed(two_joined):.4f} ms/step")
    print(f"two graphs, no join       : {timed(two_free):.4f} ms/step   (chains pipelined across steps: not a training step)")
    print(f"protein graph alone       : {timed(protein_only):.4f} ms/step")
