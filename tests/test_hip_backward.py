"""GPU parity of the hand-written BACKWARD kernels (through the nn.Module
boundary and torch.autograd): reference golden gradients, oracle autograd on
other shapes / aggregation / depth, dropout with pinned masks."""
import json
import os

import numpy as np
import pytest
import torch

import davis_synth as ds
from conftest import GOLDEN, rel_err
from gvp_hip import arena
from oracle import gvp_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


def _to(d, dev=DEV):
    return {k: (tuple(t.to(dev) for t in v) if isinstance(v, tuple) else v.to(dev)) for k, v in d.items()}


def _encoder(state=None, num_convs=2, aggr="sum", dropout=0.2, seed=None):
    from models.protein_gnn import SelectableProteinModelWrapper
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["protein_gnn_kwargs"]
    kw = dict(kw, num_convs=num_convs, aggr=aggr, dropout_rate=dropout)
    for k in ("in_channels", "edge_dim", "hidden_channels", "edge_hidden_channels"):
        kw[k] = tuple(kw[k])
    if seed is not None:
        torch.manual_seed(seed)
    m = SelectableProteinModelWrapper(**kw)
    if state is not None:
        m.load_state_dict({"gnn_model." + k: v for k, v in state.items()}, strict=False)
    return m.to(DEV)


def _check_grads(model, ref_grads, tol=2e-4):
    """max-abs error of every weight gradient relative to the largest gradient of
    its own tensor, with a floor tied to the global gradient scale (gradients that
    are analytically zero are pure rounding noise)."""
    scale = max(float(v.abs().max()) for v in ref_grads.values())
    checked = 0
    for name, p in model.gnn_model.named_parameters():
        if not p.numel():
            continue
        ref = ref_grads[name]
        err = float((p.grad.cpu() - ref).abs().max())
        assert err <= tol * float(ref.abs().max()) + 2e-6 * scale, (name, err, float(ref.abs().max()))
        checked += 1
    return checked


def test_backward_golden(lba_small, protein_params):
    """Gradients produced by the reference's own autograd (pretrained weights)."""
    g = lba_small
    model = _encoder(protein_params).eval()
    xs, xv = T(g["x_s"]).to(DEV).requires_grad_(), T(g["x_v"]).to(DEV).requires_grad_()
    out = model((xs, xv), T(g["edge_index"]).to(DEV), T(g["ntypes"]).to(DEV), T(g["etypes"]).to(DEV),
                eattr=(T(g["e_s"]).to(DEV), T(g["e_v"]).to(DEV)), batch=T(g["batch"]).to(DEV))
    assert rel_err(out, g["out"]) < 2e-5
    (out * T(g["r"]).to(DEV)).sum().backward()
    ref = {k[2:]: T(v) for k, v in g.items() if k.startswith("g_")}
    assert _check_grads(model, ref) >= 60
    assert rel_err(xs.grad, g["gin_x_s"]) < 2e-4
    assert rel_err(xv.grad, g["gin_x_v"]) < 2e-4


@pytest.mark.parametrize("fixture", ["lba_small", "lba_sparse"])
def test_edge_feature_gradients_golden(request, protein_params, fixture):
    """d(loss)/d(eattr) -- what the reference's autograd returns for the raw edge features (attribution studies; the
    training loop never asks for it) -- from the edge stage of the backward pass, against the reference's own numbers;
    the weight and node-feature gradients of the same call stay what they are without it.  lba_small takes the
    two-launch layers, lba_sparse (E <= 4 N) the fused ones."""
    g = request.getfixturevalue(fixture)
    model = _encoder(protein_params).eval()
    xs, xv = T(g["x_s"]).to(DEV).requires_grad_(), T(g["x_v"]).to(DEV).requires_grad_()
    es, ev = T(g["e_s"]).to(DEV).requires_grad_(), T(g["e_v"]).to(DEV).requires_grad_()
    out = model((xs, xv), T(g["edge_index"]).to(DEV), T(g["ntypes"]).to(DEV), T(g["etypes"]).to(DEV),
                eattr=(es, ev), batch=T(g["batch"]).to(DEV))
    (out * T(g["r"]).to(DEV)).sum().backward()
    assert es.grad.shape == es.shape and ev.grad.shape == ev.shape
    assert rel_err(es.grad, g["gin_e_s"]) < 2e-4
    assert rel_err(ev.grad, g["gin_e_v"]) < 2e-4
    assert rel_err(xs.grad, g["gin_x_s"]) < 2e-4 and rel_err(xv.grad, g["gin_x_v"]) < 2e-4
    ref = {k[2:]: T(v) for k, v in g.items() if k.startswith("g_")}
    assert _check_grads(model, ref) >= 60
    # only the edge features ask for a gradient (frozen weights, no node-feature gradient): same numbers
    frozen = _encoder(protein_params).eval().requires_grad_(False)
    es2 = T(g["e_s"]).to(DEV).requires_grad_()
    out2 = frozen((T(g["x_s"]).to(DEV), T(g["x_v"]).to(DEV)), T(g["edge_index"]).to(DEV), T(g["ntypes"]).to(DEV),
                  T(g["etypes"]).to(DEV), eattr=(es2, T(g["e_v"]).to(DEV)), batch=T(g["batch"]).to(DEV))
    (out2 * T(g["r"]).to(DEV)).sum().backward()
    assert rel_err(es2.grad, es.grad) < 1e-5          # (not bitwise: d h[src] is accumulated with float atomics)


@pytest.mark.parametrize("case", ["c1_davis16_sum", "knn_mean", "radius_mean", "depth4_ragged"])
def test_backward_vs_oracle_autograd(protein_params, case):
    if case == "c1_davis16_sum":
        gb, nc, aggr, state = ds.protein_batch(16, 1), 2, "sum", protein_params
    elif case == "knn_mean":
        gb, nc, aggr, state = ds.protein_batch(2, 2, length=150, thresh=20, thresh_type="num"), 2, "mean", protein_params
    elif case == "radius_mean":                # sparse graph: the fused conv + node-update launch, with aggr='mean'
        gb, nc, aggr, state = ds.protein_batch(3, 7, lengths=[40, 77, 120]), 2, "mean", protein_params
    else:
        gb, nc, aggr, state = ds.protein_batch(4, 3, lengths=[1, 17, 64, 33], thresh=7.0), 4, "sum", None
    model = _encoder(state, nc, aggr, seed=5).eval()
    d = ds.to_torch(gb)
    P = {k: v.detach().cpu().clone().requires_grad_(v.numel() > 0) for k, v in model.gnn_model.state_dict().items()}
    xs, xv = d["x"][0].clone().requires_grad_(), d["x"][1].clone().requires_grad_()
    ref_out = O.protein_lba_forward(P, (xs, xv), d["edge_index"], d["ntypes"], d["etypes"], d["eattr"],
                                    num_convs=nc, aggr=aggr)
    r = torch.randn(ref_out.shape, generator=torch.Generator().manual_seed(3))
    (ref_out * r).sum().backward()
    dd = _to(d)
    gxs, gxv = dd["x"][0].clone().requires_grad_(), dd["x"][1].clone().requires_grad_()
    out = model((gxs, gxv), dd["edge_index"], dd["ntypes"], dd["etypes"], eattr=dd["eattr"])
    assert rel_err(out, ref_out) < 2e-5
    (out * r.to(DEV)).sum().backward()
    _check_grads(model, {k: v.grad for k, v in P.items() if v.numel()})
    assert rel_err(gxs.grad, xs.grad) < 2e-4 and rel_err(gxv.grad, xv.grad) < 2e-4


def test_weights_only_and_accumulation(protein_params):
    """No input gradients requested; .grad accumulates over two backward passes."""
    model = _encoder(protein_params).eval()
    d = _to(ds.to_torch(ds.protein_batch(3, 8, length=60)))
    for _ in range(2):
        model(d["x"], d["edge_index"], d["ntypes"], d["etypes"], eattr=d["eattr"]).square().mean().backward()
    g2 = {n: p.grad.clone() for n, p in model.gnn_model.named_parameters() if p.numel()}
    model.zero_grad()
    model(d["x"], d["edge_index"], d["ntypes"], d["etypes"], eattr=d["eattr"]).square().mean().backward()
    # (relative to the tensor's own maximum, with a floor tied to the global gradient scale: gradients that are analytically
    # zero -- gvp_edge.0.wsv at ~1e-11 -- are rounding noise that the d h[src] float atomics reorder from run to run)
    scale = max(float(p.grad.abs().max()) for p in model.gnn_model.parameters() if p.numel())
    for n, p in model.gnn_model.named_parameters():
        if p.numel():
            err = float((g2[n] - 2 * p.grad).abs().max())
            assert err <= 1e-5 * float((2 * p.grad).abs().max()) + 1e-7 * scale, (n, err)


def test_training_dropout_with_pinned_masks(protein_params, monkeypatch):
    """Training mode: the kernels apply the dropout masks they are given (drawn by
    torch) exactly like gvp_layers.Dropout; forward and gradients match the oracle
    run with the same masks."""
    from gvp_hip import autograd_ops
    gb = ds.protein_batch(4, 11, length=50)
    N = gb.num_nodes
    gen = torch.Generator().manual_seed(0)
    pinned = [((torch.rand(N, 20, generator=gen) < 0.8).float() / 0.8) for _ in range(4)]
    monkeypatch.setattr(autograd_ops, "PINNED_MASKS", lambda count, n, width, p, dev: torch.stack(pinned).to(dev))
    model = _encoder(protein_params).train()
    d = ds.to_torch(gb)
    dd = _to(d)
    out = model(dd["x"], dd["edge_index"], dd["ntypes"], dd["etypes"], eattr=dd["eattr"])
    r = torch.randn(out.shape, generator=torch.Generator().manual_seed(4))
    (out * r.to(DEV)).sum().backward()
    P = {k: v.detach().cpu().clone().requires_grad_(v.numel() > 0) for k, v in model.gnn_model.state_dict().items()}
    ref = O.protein_lba_forward(P, d["x"], d["edge_index"], d["ntypes"], d["etypes"], d["eattr"],
                                masks=[(pinned[0], pinned[1]), (pinned[2], pinned[3])])
    assert rel_err(out, ref) < 2e-5
    (ref * r).sum().backward()
    _check_grads(model, {k: v.grad for k, v in P.items() if v.numel()})


def test_in_kernel_dropout_statistics_and_determinism():
    """The factors the kernels generate (Philox keyed by seed / offset / mask id / node / channel): 0 or 1/(1-p),
    drop rate p, independent across masks, nodes and channels, a pure function of the seed."""
    from gvp_hip import autograd_ops, ops
    torch.manual_seed(7)
    seed = autograd_ops.draw_seed(DEV)
    m = ops.dropout_masks(seed, 0.2, 0, 4, 20000, 20)
    assert m.shape == (4, 20000, 20)
    assert set(np.unique(m.cpu().numpy()).round(4)) == {0.0, 1.25}
    drop = (m == 0).float()
    assert abs(float(drop.mean()) - 0.2) < 0.005
    assert float((drop.mean(dim=(1, 2)) - 0.2).abs().max()) < 0.01          # every mask
    assert float((drop.mean(dim=(0, 1)) - 0.2).abs().max()) < 0.01          # every channel
    a, b = drop[0].flatten(), drop[1].flatten()                               # two masks of one step: uncorrelated
    assert abs(float(((a - a.mean()) * (b - b.mean())).mean())) < 0.005
    assert abs(float(((drop[0, :-1] - 0.2) * (drop[0, 1:] - 0.2)).mean())) < 0.005   # neighbouring nodes
    assert torch.equal(m, ops.dropout_masks(seed, 0.2, 0, 4, 20000, 20))      # deterministic in the seed
    assert torch.equal(m[2:], ops.dropout_masks(seed, 0.2, 2, 2, 20000, 20))  # mask id = stream + index
    assert not torch.equal(m, ops.dropout_masks(seed + 1, 0.2, 0, 4, 20000, 20))
    # the same generator compiled with g++ (pinned to the published Philox4x32-10 known answers in
    # tests/test_host_math.py) yields the same factors: the GPU masks are exactly that function
    import ctypes, subprocess, tempfile
    from conftest import REPO
    so = os.path.join(tempfile.mkdtemp(), "host_rng.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(REPO, "tests", "host_math", "host_rng.cpp")])
    host = np.zeros((3000, 20), np.float32)
    sv = [int(v) for v in seed.cpu()]
    ctypes.CDLL(so).host_dropout_mask(ctypes.c_ulonglong(sv[0]), ctypes.c_ulonglong(sv[1]), 1, ctypes.c_longlong(3000), 20,
                                      ctypes.c_float(0.2), host.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(host, m[1, :3000].cpu().numpy())
    g = ops.dropout_masks(seed, 0.5, 0, 1, 5000, 64)                          # GINE row width
    assert set(np.unique(g.cpu().numpy()).round(4)) == {0.0, 2.0} and abs(float((g == 0).float().mean()) - 0.5) < 0.01


def test_training_step_with_in_kernel_dropout_matches_oracle(protein_params, molecule_params):
    """Production training mode (masks generated INSIDE the forward kernels and regenerated inside the backward
    kernels): export the factors for the step's seed and run the oracle with exactly those -- forward and every
    gradient of both encoders must agree."""
    from gvp_hip import autograd_ops, ops
    from models.molecule_gnn import SelectableMoleculeModelWrapper
    pb, mb = ds.pair_batch(4, 13, lengths=[50, 37, 64, 45])
    pd, md = ds.to_torch(pb), ds.to_torch(mb)
    # ---- protein
    model = _encoder(protein_params).train()
    dd = _to(pd)
    out = model(dd["x"], dd["edge_index"], dd["ntypes"], dd["etypes"], eattr=dd["eattr"])
    seed = autograd_ops.last_seed("lba")                   # {seed, offset} of this pass, written by its first kernel
    masks = ops.dropout_masks(seed, 0.2, 0, 4, pb.num_nodes, 20).cpu()
    r = torch.randn(out.shape, generator=torch.Generator().manual_seed(4))
    (out * r.to(DEV)).sum().backward()
    P = {k: v.detach().cpu().clone().requires_grad_(v.numel() > 0) for k, v in model.gnn_model.state_dict().items()}
    ref = O.protein_lba_forward(P, pd["x"], pd["edge_index"], pd["ntypes"], pd["etypes"], pd["eattr"],
                                masks=[(masks[0], masks[1]), (masks[2], masks[3])])
    assert rel_err(out, ref) < 2e-5
    (ref * r).sum().backward()
    _check_grads(model, {k: v.grad for k, v in P.items() if v.numel()})
    ev = model.eval()(dd["x"], dd["edge_index"], dd["ntypes"], dd["etypes"], eattr=dd["eattr"])
    assert rel_err(ev, ref) > 1e-2                          # dropout really was applied
    # ---- drug (dropout between the two GINE layers, molecule_gnn.py:262)
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["molecule_gnn_kwargs"]
    mol = SelectableMoleculeModelWrapper(**kw)
    mol.load_state_dict({"gnn_model." + k: v for k, v in molecule_params.items()})
    mol = mol.to(DEV).train()
    dm = _to(md)
    gx = dm["x"].clone().requires_grad_()
    mout = mol(gx, dm["edge_index"], dm["ntypes"], dm["etypes"], eattr=dm["eattr"])
    mseed = autograd_ops.last_seed("gine")
    mmask = ops.dropout_masks(mseed, 0.2, 0, 1, mb.num_nodes, 16).cpu()
    Q = {k: v.clone().requires_grad_(True) for k, v in molecule_params.items()}
    xr = md["x"].clone().requires_grad_()
    mref = O.molecule_gine_forward(Q, xr, md["edge_index"], md["ntypes"], md["etypes"], md["eattr"], masks=[mmask[0]])
    assert rel_err(mout, mref) < 2e-5
    r2 = torch.randn(mref.shape, generator=torch.Generator().manual_seed(5))
    (mref * r2).sum().backward()
    (mout * r2.to(DEV)).sum().backward()
    for name, p in mol.gnn_model.named_parameters():
        assert rel_err(p.grad, Q[name].grad) < 2e-4, name
    assert rel_err(gx.grad, xr.grad) < 2e-4


def test_adam_step_moves_the_arena(protein_params):
    """An optimizer step through the ordinary nn.Module API changes what the kernels compute."""
    model = _encoder(protein_params).eval()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    d = _to(ds.protein_batch(2, 2, length=40) and ds.to_torch(ds.protein_batch(2, 2, length=40)))
    before = model(d["x"], d["edge_index"], d["ntypes"], d["etypes"], eattr=d["eattr"]).detach().clone()
    loss = model(d["x"], d["edge_index"], d["ntypes"], d["etypes"], eattr=d["eattr"]).square().mean()
    loss.backward()
    opt.step()
    after = model(d["x"], d["edge_index"], d["ntypes"], d["etypes"], eattr=d["eattr"]).detach()
    assert float((after - before).abs().max()) > 1e-4
    assert float(after.square().mean()) < float(before.square().mean())


def test_gine_backward_vs_oracle(molecule_params):
    """Drug encoder gradients (weights incl. eps, and atom-feature inputs) vs oracle autograd."""
    from models.molecule_gnn import SelectableMoleculeModelWrapper
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["molecule_gnn_kwargs"]
    model = SelectableMoleculeModelWrapper(**kw)
    model.load_state_dict({"gnn_model." + k: v for k, v in molecule_params.items()})
    model = model.to(DEV).eval()
    d = ds.to_torch(ds.drug_batch(12, 3))
    P = {k: v.clone().requires_grad_(True) for k, v in molecule_params.items()}
    x = d["x"].clone().requires_grad_()
    ref = O.molecule_gine_forward(P, x, d["edge_index"], d["ntypes"], d["etypes"], d["eattr"])
    r = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1))
    (ref * r).sum().backward()
    dd = _to(d)
    gx = dd["x"].clone().requires_grad_()
    out = model(gx, dd["edge_index"], dd["ntypes"], dd["etypes"], eattr=dd["eattr"])
    assert rel_err(out, ref) < 2e-5
    (out * r.to(DEV)).sum().backward()
    for name, p in model.gnn_model.named_parameters():
        assert rel_err(p.grad, P[name].grad) < 2e-4, name
    assert rel_err(gx.grad, x.grad) < 2e-4


def test_gine_backward_dense_graph(molecule_params):
    """A graph no molecule looks like: 8-12 incoming edges per atom (more than 64 edges per 16-atom tile:
    the multi-chunk path of the GINE backward), a few atoms without edges, a ragged last tile."""
    from models.molecule_gnn import SelectableMoleculeModelWrapper
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["molecule_gnn_kwargs"]
    model = SelectableMoleculeModelWrapper(**kw)
    model.load_state_dict({"gnn_model." + k: v for k, v in molecule_params.items()})
    model = model.to(DEV).eval()
    base = ds.to_torch(ds.drug_batch(5, 7))
    gen = torch.Generator().manual_seed(5)
    n = base["x"].shape[0] - 3                       # not a multiple of 16 for this seed's sizes
    x, ntypes = base["x"][:n].clone(), base["ntypes"][:n].clone()
    deg = torch.randint(8, 13, (n,), generator=gen)
    deg[torch.randperm(n, generator=gen)[:4]] = 0     # isolated atoms
    dst = torch.repeat_interleave(torch.arange(n), deg)
    src = torch.randint(0, n, (int(deg.sum()),), generator=gen)
    perm = torch.randperm(dst.numel(), generator=gen)
    edge_index = torch.stack([src, dst])[:, perm].contiguous()
    E = edge_index.shape[1]
    etypes = torch.randint(0, 5, (E,), generator=gen)
    eattr = torch.randn(E, base["eattr"].shape[1], generator=gen)
    P = {k: v.clone().requires_grad_(True) for k, v in molecule_params.items()}
    xr = x.clone().requires_grad_()
    ref = O.molecule_gine_forward(P, xr, edge_index, ntypes, etypes, eattr)
    r = torch.randn(ref.shape, generator=torch.Generator().manual_seed(2))
    (ref * r).sum().backward()
    gx = x.to(DEV).requires_grad_()
    out = model(gx, edge_index.to(DEV), ntypes.to(DEV), etypes.to(DEV), eattr=eattr.to(DEV))
    assert rel_err(out, ref) < 2e-5
    (out * r.to(DEV)).sum().backward()
    for name, p in model.gnn_model.named_parameters():
        assert rel_err(p.grad, P[name].grad) < 2e-4, name
    assert rel_err(gx.grad, xr.grad) < 2e-4


def test_joint_model_training_step(pretrained):
    """train_model.py:548-587 in miniature: JointGNN in train() mode (dropout on),
    MSE loss, backward through head (torch) + both encoders (HIP), Adam step; the
    loss goes down and every parameter that should get a gradient gets one."""
    from models.joint_gnn import JointGNN
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
    model = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"],
                     **kw["joint_gnn_kwargs"])
    model.load_state_dict(pretrained, strict=True)
    model = model.to(DEV).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    p, m = ds.pair_batch(6, 2, lengths=[40, 60, 33, 80, 50, 45])
    pd, md = _to(ds.to_torch(p)), _to(ds.to_torch(m))
    target = torch.linspace(-1, 1, 6, device=DEV).unsqueeze(-1)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        pred, _ = model(pd, md)
        loss = torch.nn.functional.mse_loss(pred, target)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and np.mean(losses[-3:]) < np.mean(losses[:3])
    missing = [n for n, q in model.named_parameters() if q.numel() and q.grad is None]
    assert not missing, missing


def test_large_batch_linearity(protein_params):
    """Full-size property (no oracle needed): 240 x 300 residues = 72,000 nodes /
    215k edges exercises the global-memory CSR scan (N > 36k), the persistent
    multi-tile loops of the backward kernels (> 4096 waves' worth of tiles) and the
    private read-add-store accumulation.  Graphs are independent, so the output of
    the big batch equals the outputs of its halves and every weight gradient equals
    the SUM of the halves' gradients."""
    model = _encoder(protein_params).eval()
    halves = [ds.protein_batch(120, 31), ds.protein_batch(120, 32)]
    big = ds.collate([dict(x_s=h.x_s[h.ptr[i]:h.ptr[i + 1]], x_v=h.x_v[h.ptr[i]:h.ptr[i + 1]],
                           edge_index=h.edge_index[:, h.eptr[i]:h.eptr[i + 1]] - h.ptr[i],
                           e_s=h.e_s[h.eptr[i]:h.eptr[i + 1]], e_v=h.e_v[h.eptr[i]:h.eptr[i + 1]],
                           ntypes=h.ntypes[h.ptr[i]:h.ptr[i + 1]], etypes=h.etypes[h.eptr[i]:h.eptr[i + 1]])
                      for h in halves for i in range(h.num_graphs)])
    assert big.num_nodes == 72000
    gen = torch.Generator(device=DEV).manual_seed(0)
    r = torch.randn(big.num_nodes, 64, device=DEV, generator=gen)
    params = [p for p in model.parameters() if p.numel()]

    def run(gb, rr):
        d = _to(ds.to_torch(gb))
        out = model(d["x"], d["edge_index"], d["ntypes"], d["etypes"], eattr=d["eattr"])
        return out.detach(), torch.autograd.grad(out, params, rr)

    out_big, g_big = run(big, r)
    n0 = halves[0].num_nodes
    out_a, g_a = run(halves[0], r[:n0])
    out_b, g_b = run(halves[1], r[n0:])
    assert torch.isfinite(out_big).all()
    assert rel_err(out_big, torch.cat([out_a, out_b])) < 1e-6      # same kernels, same per-node arithmetic
    scale = max(float(g.abs().max()) for g in g_big)
    for gb_, ga_, gb2_ in zip(g_big, g_a, g_b):
        ref = ga_ + gb2_
        assert float((gb_ - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-6 * scale


def test_backward_without_edges(protein_params, molecule_params):
    """Degenerate batches the reference handles (nothing to aggregate): graphs with no edges at all, a single
    node.  Forward and every gradient against the oracle; exercises the empty-segment paths of every kernel."""
    from models.molecule_gnn import SelectableMoleculeModelWrapper
    # protein: 5 isolated residues
    gb = ds.protein_batch(1, 9, length=5)
    d = ds.to_torch(gb)
    keep = torch.zeros(d["edge_index"].shape[1], dtype=torch.bool)
    d = dict(d, edge_index=d["edge_index"][:, keep], etypes=d["etypes"][keep],
             eattr=(d["eattr"][0][keep], d["eattr"][1][keep]))
    P = {k: v.clone().requires_grad_(True) for k, v in protein_params.items()}
    ref = O.protein_lba_forward(P, d["x"], d["edge_index"], d["ntypes"], d["etypes"], d["eattr"])
    r = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3))
    (ref * r).sum().backward()
    model = _encoder(protein_params).eval()
    dd = _to(d)
    out = model(dd["x"], dd["edge_index"], dd["ntypes"], dd["etypes"], eattr=dd["eattr"])
    assert rel_err(out, ref) < 2e-5
    (out * r.to(DEV)).sum().backward()
    assert _check_grads(model, {k: v.grad for k, v in P.items() if v.grad is not None}) >= 40
    # drug: one molecule's atoms, no bonds
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["molecule_gnn_kwargs"]
    mol = SelectableMoleculeModelWrapper(**kw)
    mol.load_state_dict({"gnn_model." + k: v for k, v in molecule_params.items()})
    mol = mol.to(DEV).eval()
    m = ds.to_torch(ds.drug_batch(1, 4))
    ei = m["edge_index"][:, :0]
    et, ea = m["etypes"][:0], m["eattr"][:0]
    Q = {k: v.clone().requires_grad_(True) for k, v in molecule_params.items()}
    xr = m["x"].clone().requires_grad_()
    mref = O.molecule_gine_forward(Q, xr, ei, m["ntypes"], et, ea)
    r2 = torch.randn(mref.shape, generator=torch.Generator().manual_seed(4))
    (mref * r2).sum().backward()
    gx = m["x"].to(DEV).requires_grad_()
    mout = mol(gx, ei.to(DEV), m["ntypes"].to(DEV), et.to(DEV), eattr=ea.to(DEV))
    assert rel_err(mout, mref) < 2e-5
    (mout * r2.to(DEV)).sum().backward()
    for name, p in mol.gnn_model.named_parameters():
        g = Q[name].grad
        if g is None or float(g.abs().max()) == 0.0:        # lin.* gets no gradient without edges
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
        else:
            assert rel_err(p.grad, g) < 2e-4, name
    assert rel_err(gx.grad, xr.grad) < 2e-4



def test_davis_b64_both_encoders_fwd_bwd_vs_oracle(protein_params, molecule_params):
    """BASELINE config 2 at EXACTLY its size -- the batch bench.py times (64 pairs x 300 residues, 4 A radius graphs,
    ~40-atom drugs, seed 0, CASTER-DTA(2,2), train mode, dropout 0.2) -- both encoders forward + every gradient against the
    CPU oracle run with the masks the kernels drew (until now this size was covered by properties and
    self-comparison only).  Yardstick: the fp64 oracle, and what a gradient at this size CAN agree to: a weight gradient here
    sums over 19,200 residues / 57k edges = ~7e6 ReLU / norm-clamp decisions, and a pre-activation within rounding of zero
    takes the other branch.  Measured on the CPU (fp64 oracle, this batch): the reference-order fp32 evaluation differs from
    fp64 by 7e-7 (median over tensors) but 9e-4 at the 90th percentile and 2.7e-3 at worst in relative L2; a 1e-6 relative
    perturbation of the weights -- the kernels' rounding level -- moves the fp64 gradients by 2-3e-3 (median) and 2.7e-2
    (worst).  So: forward within 2e-5; every gradient tensor within 3e-2 in relative L2, their median within 5e-3.  (Exact
    gradient parity, 2e-4 per element, is what the smaller oracle / golden cases check, where no decision sits on the edge.)"""
    from gvp_hip import autograd_ops, ops
    from models.molecule_gnn import SelectableMoleculeModelWrapper
    pb, mb = ds.pair_batch(64, 0, length=300, thresh=4.0, thresh_type="dist")
    assert pb.num_nodes == 19200
    pd, md = ds.to_torch(pb), ds.to_torch(mb)
    model = _encoder(protein_params).train()
    dd = _to(pd)
    out = model(dd["x"], dd["edge_index"], dd["ntypes"], dd["etypes"], eattr=dd["eattr"])
    seed = autograd_ops.last_seed("lba")
    masks = ops.dropout_masks(seed, 0.2, 0, 4, pb.num_nodes, 20).cpu()
    r = torch.randn(out.shape, generator=torch.Generator().manual_seed(4))
    (out * r.to(DEV)).sum().backward()

    def oracle(dt):
        cast = lambda t: t.to(dt) if t.is_floating_point() else t
        P = {k: cast(v.detach().cpu().clone()).requires_grad_(v.numel() > 0) for k, v in model.gnn_model.state_dict().items()}
        ref = O.protein_lba_forward(P, tuple(cast(t) for t in pd["x"]), pd["edge_index"], pd["ntypes"], pd["etypes"],
                                    tuple(cast(t) for t in pd["eattr"]),
                                    masks=[(cast(masks[0]), cast(masks[1])), (cast(masks[2]), cast(masks[3]))])
        (ref * cast(r)).sum().backward()
        return ref.detach(), {k: v.grad for k, v in P.items() if v.numel()}
    ref64, g64 = oracle(torch.float64)
    assert rel_err(out, ref64) < 2e-5
    scale = max(float(v.abs().max()) for v in g64.values())
    checked, rels = 0, []
    for name, p in model.gnn_model.named_parameters():
        if not p.numel():
            continue
        diff = p.grad.cpu().double() - g64[name]
        rels.append(float(diff.norm()) / (float(g64[name].norm()) + 1e-6 * scale))
        assert rels[-1] <= 3e-2, (name, rels[-1])
        checked += 1
    assert checked >= 60 and sorted(rels)[len(rels) // 2] <= 5e-3, sorted(rels)[len(rels) // 2]
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["molecule_gnn_kwargs"]
    mol = SelectableMoleculeModelWrapper(**kw)
    mol.load_state_dict({"gnn_model." + k: v for k, v in molecule_params.items()})
    mol = mol.to(DEV).train()
    dm = _to(md)
    mout = mol(dm["x"], dm["edge_index"], dm["ntypes"], dm["etypes"], eattr=dm["eattr"])
    mmask = ops.dropout_masks(autograd_ops.last_seed("gine"), 0.2, 0, 1, mb.num_nodes, 16).cpu()
    r2 = torch.randn(mout.shape, generator=torch.Generator().manual_seed(5))
    (mout * r2.to(DEV)).sum().backward()

    def moracle(dt):
        Q = {k: v.clone().to(dt).requires_grad_(True) for k, v in molecule_params.items()}
        mref = O.molecule_gine_forward(Q, md["x"].to(dt), md["edge_index"], md["ntypes"], md["etypes"], md["eattr"].to(dt),
                                       masks=[mmask[0].to(dt)])
        (mref * r2.to(dt)).sum().backward()
        return mref.detach(), {k: v.grad for k, v in Q.items()}
    m64, q64 = moracle(torch.float64)
    assert rel_err(mout, m64) < 2e-5
    mscale = max(float(v.abs().max()) for v in q64.values())
    for name, p in mol.gnn_model.named_parameters():
        diff = p.grad.cpu().double() - q64[name]
        assert float(diff.norm()) <= 3e-2 * (float(q64[name].norm()) + 1e-6 * mscale), (name, float(diff.norm()))


def test_in_place_edits_between_forward_and_backward_raise(protein_params, molecule_params):
    """the C++ autograd nodes re-read inputs (and the drug encoder its weights) at backward time; an in-place
    write in between must raise like stock autograd's saved-tensor version check, and create_graph=True must not return
    silently non-differentiable gradients."""
    from models.molecule_gnn import SelectableMoleculeModelWrapper
    from gvp_hip import _lib
    if _lib.bridge() is None:
        pytest.skip("C++ bridge not in use (CGVP_BRIDGE=0 / CGVP_LIB_PATH)")
    pb, mb = ds.pair_batch(2, 3, lengths=[40, 33])
    dd, dm = _to(ds.to_torch(pb)), _to(ds.to_torch(mb))
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["molecule_gnn_kwargs"]
    mol = SelectableMoleculeModelWrapper(**kw)
    mol.load_state_dict({"gnn_model." + k: v for k, v in molecule_params.items()})
    mol = mol.to(DEV).eval()
    out = mol(dm["x"], dm["edge_index"], dm["ntypes"], dm["etypes"], eattr=dm["eattr"])
    with torch.no_grad():
        next(p for p in mol.parameters() if p.numel()).mul_(1.5)           # an optimizer step before this pass's backward
    with pytest.raises(RuntimeError, match="modified by an in-place"):
        out.sum().backward()
    model = _encoder(protein_params).eval()
    xs = dd["x"][0].clone().requires_grad_()
    xin = xs * 1.0                                                          # (a non-leaf the test may write in place)
    out = model((xin, dd["x"][1]), dd["edge_index"], dd["ntypes"], dd["etypes"], eattr=dd["eattr"])
    with torch.no_grad():
        xin.add_(1.0)
    with pytest.raises(RuntimeError, match="modified by an in-place"):
        out.sum().backward()
    out = model(dd["x"], dd["edge_index"], dd["ntypes"], dd["etypes"], eattr=dd["eattr"])
    w = [p for p in model.parameters() if p.numel()]
    with pytest.raises(RuntimeError, match="not differentiable twice"):
        torch.autograd.grad(out.sum(), w, create_graph=True)
    out = model(dd["x"], dd["edge_index"], dd["ntypes"], dd["etypes"], eattr=dd["eattr"])       # and the normal path still runs
    out.sum().backward()
