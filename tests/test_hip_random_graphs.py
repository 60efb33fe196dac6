"""GPU parity on RANDOM graphs (arbitrary COO lists, not geometry): isolated residues, duplicate edges, self loops, hubs,
node counts that are not multiples of the 16-item tile, unsorted edge order -- forward and every gradient of both
encoders against the CPU oracle.  Complements the geometric batches of the other parity tests."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from oracle import gvp_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _random_coo(rng, n, e, hubs):
    src = rng.integers(0, n, size=e)
    dst = rng.integers(0, n, size=e)
    if hubs and n > 4 and e > 8:                      # a few targets collect a third of the edges (segments longer than a tile)
        hub = rng.integers(0, n, size=2)
        pick = rng.random(e) < 0.33
        dst[pick] = hub[rng.integers(0, 2, size=int(pick.sum()))]
    if e > 3:
        src[:2] = dst[:2]                             # self loops
        src[2], dst[2] = src[3], dst[3]               # a duplicate edge
    return torch.from_numpy(np.stack([src, dst]).astype(np.int64))


def _protein_inputs(rng, n, e, hubs):
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    x_s, x_v = torch.randn(n, 17, generator=g), torch.randn(n, 3, 3, generator=g)
    if n > 2:
        x_v[1] = 0                                    # zero vectors: the clamped norms
    return dict(x=(x_s, x_v), edge_index=_random_coo(rng, n, e, hubs), ntypes=torch.randint(0, 20, (n,), generator=g),
                etypes=torch.zeros(e, dtype=torch.int64), eattr=(torch.randn(e, 32, generator=g), torch.randn(e, 1, 3, generator=g)))


CASES = [(1, 0, 2, "sum"), (1, 1, 2, "sum"), (15, 40, 2, "mean"), (17, 3, 1, "sum"), (33, 500, 2, "sum"), (100, 100, 3, "mean"),
         (257, 1000, 2, "sum"), (64, 0, 2, "mean"), (200, 3000, 2, "mean"), (47, 188, 4, "sum")]


@pytest.mark.parametrize("n,e,nc,aggr", CASES)
def test_protein_random_graph_vs_oracle(n, e, nc, aggr):
    from models.protein_gnn import SelectableProteinModelWrapper
    rng = np.random.default_rng(1000 * n + e)
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["protein_gnn_kwargs"]
    kw = dict(kw, num_convs=nc, aggr=aggr)
    for k in ("in_channels", "edge_dim", "hidden_channels", "edge_hidden_channels"):
        kw[k] = tuple(kw[k])
    torch.manual_seed(n + e)
    model = SelectableProteinModelWrapper(**kw).to(DEV).eval()
    d = _protein_inputs(rng, n, e, hubs=e >= 500)
    P = {k: v.detach().cpu().clone().requires_grad_(v.numel() > 0) for k, v in model.gnn_model.state_dict().items()}
    xs, xv = d["x"][0].clone().requires_grad_(), d["x"][1].clone().requires_grad_()
    ref = O.protein_lba_forward(P, (xs, xv), d["edge_index"], d["ntypes"], d["etypes"], d["eattr"], num_convs=nc, aggr=aggr)
    r = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3))
    (ref * r).sum().backward()
    to = lambda t: t.to(DEV)
    gxs, gxv = to(d["x"][0]).requires_grad_(), to(d["x"][1]).requires_grad_()
    out = model((gxs, gxv), to(d["edge_index"]), to(d["ntypes"]), to(d["etypes"]), eattr=tuple(to(t) for t in d["eattr"]))
    assert rel_err(out, ref) < 2e-5
    (out * r.to(DEV)).sum().backward()
    scale = max(float(v.grad.abs().max()) for v in P.values() if v.grad is not None)
    for name, p in model.gnn_model.named_parameters():
        if p.numel():
            want = P[name].grad
            assert float((p.grad.cpu() - want).abs().max()) <= 2e-4 * float(want.abs().max()) + 2e-6 * scale, name
    assert rel_err(gxs.grad, xs.grad) < 2e-4 and rel_err(gxv.grad, xv.grad) < 2e-4


@pytest.mark.parametrize("n,e", [(1, 0), (3, 2), (16, 34), (41, 90), (130, 600), (64, 0)])
def test_drug_random_graph_vs_oracle(molecule_params, n, e):
    from models.molecule_gnn import SelectableMoleculeModelWrapper
    rng = np.random.default_rng(77 * n + e)
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["molecule_gnn_kwargs"]
    model = SelectableMoleculeModelWrapper(**kw)
    model.load_state_dict({"gnn_model." + k: v for k, v in molecule_params.items()})
    model = model.to(DEV).eval()
    g = torch.Generator().manual_seed(n + e)
    x, ea = torch.randn(n, 41, generator=g), torch.randn(e, 9, generator=g)
    nt, et = torch.randint(0, 11, (n,), generator=g), torch.randint(0, 5, (e,), generator=g)
    ei = _random_coo(rng, n, e, hubs=e >= 500)
    P = {k: v.clone().requires_grad_(True) for k, v in molecule_params.items()}
    xr = x.clone().requires_grad_()
    ref = O.molecule_gine_forward(P, xr, ei, nt, et, ea)
    r = torch.randn(ref.shape, generator=torch.Generator().manual_seed(5))
    (ref * r).sum().backward()
    xg = x.to(DEV).requires_grad_()
    out = model(xg, ei.to(DEV), nt.to(DEV), et.to(DEV), eattr=ea.to(DEV))
    assert rel_err(out, ref) < 2e-5
    (out * r.to(DEV)).sum().backward()
    scale = max(float(v.grad.abs().max()) for v in P.values())
    for name, p in model.gnn_model.named_parameters():
        want = P[name].grad
        assert float((p.grad.cpu() - want).abs().max()) <= 2e-4 * float(want.abs().max()) + 2e-6 * scale, name
    assert rel_err(xg.grad, xr.grad) < 2e-4


def test_out_of_range_edges_are_dropped_everywhere():
    """cgvp_csr_from_coo drops edges whose endpoints are out of range (the reference's index_select would raise; a GPU
    kernel must never fault on them).  Forward and every gradient then equal the run on the cleaned edge list -- in
    particular the edge-embedding backward, which walks all E sorted positions, skips the unused tail."""
    from models.protein_gnn import SelectableProteinModelWrapper
    rng = np.random.default_rng(5)
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["protein_gnn_kwargs"]
    for k in ("in_channels", "edge_dim", "hidden_channels", "edge_hidden_channels"):
        kw[k] = tuple(kw[k])
    torch.manual_seed(9)
    model = SelectableProteinModelWrapper(**kw).to(DEV).eval()
    params = [p for p in model.parameters() if p.numel()]
    n, e = 70, 230
    d = _protein_inputs(rng, n, e, hubs=False)
    ei = d["edge_index"].clone()
    bad = torch.tensor([3, 57, 111, 229])
    ei[0, bad[:2]] = torch.tensor([n, -1])                 # source out of range
    ei[1, bad[2:]] = torch.tensor([n + 5, -7])             # target out of range
    keep = torch.ones(e, dtype=torch.bool)
    keep[bad] = False
    to = lambda t: t.to(DEV)
    r = torch.randn(n, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))

    def run(edge_index, etypes, eattr):
        out = model((to(d["x"][0]), to(d["x"][1])), to(edge_index), to(d["ntypes"]), to(etypes), eattr=tuple(to(t) for t in eattr))
        return out.detach(), torch.autograd.grad(out, params, r)

    out_bad, g_bad = run(ei, d["etypes"], d["eattr"])
    out_ok, g_ok = run(d["edge_index"][:, keep], d["etypes"][keep], tuple(t[keep] for t in d["eattr"]))
    assert torch.isfinite(out_bad).all() and rel_err(out_bad, out_ok) < 1e-6
    scale = max(float(g.abs().max()) for g in g_ok)
    for a, b in zip(g_bad, g_ok):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 2e-6 * scale
