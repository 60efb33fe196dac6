"""GPU parity at the FULL size of the BASELINE.json configurations that round 1 left untested, and the
reference's own numbers sent straight through the production (one-launch-per-layer) kernels.

  * lba_sparse golden  -- reference outputs + reference-autograd gradients on a 4 A radius batch (E <= 4N),
                          through cgvp_conv_layer_fwd and the backward behind it (no stages, no unfused path)
  * C4                 -- ONE 1,000-residue kNN-20 graph (~20k edges), forward + every gradient vs the oracle
  * C3                 -- 32 pairs with lengths drawn from the KIBA sequence-length table (N ~ 23k):
                          oracle on a 4-pair subset, batch == per-graph, gradient linearity over a split
"""
import json
import os

import numpy as np
import pytest
import torch

import davis_synth as ds
from conftest import GOLDEN, rel_err
from gvp_hip import _lib, ops
from oracle import gvp_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


def _to(d, dev=DEV):
    return {k: (tuple(t.to(dev) for t in v) if isinstance(v, tuple) else v.to(dev)) for k, v in d.items()}


def _protein(state):
    from models.protein_gnn import SelectableProteinModelWrapper
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["protein_gnn_kwargs"]
    for k in ("in_channels", "edge_dim", "hidden_channels", "edge_hidden_channels"):
        kw[k] = tuple(kw[k])
    m = SelectableProteinModelWrapper(**kw)
    m.load_state_dict({"gnn_model." + k: v for k, v in state.items()})
    return m.to(DEV).eval()


def _molecule(state):
    from models.molecule_gnn import SelectableMoleculeModelWrapper
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["molecule_gnn_kwargs"]
    m = SelectableMoleculeModelWrapper(**kw)
    m.load_state_dict({"gnn_model." + k: v for k, v in state.items()})
    return m.to(DEV).eval()


def _plan(N, E, num_convs=2, prebuilt=0):
    """cgvp_lba_forward_plan: (launches, fused?) of the whole-pass entry point for these sizes (host-only)."""
    import ctypes as C
    fused = C.c_int32(-1)
    n = _lib.lib().cgvp_lba_forward_plan(N, E, num_convs, prebuilt, 0 if ops.FUSE_LAYER else 1, C.byref(fused))
    return n, fused.value


def _check_grads(model, ref_grads, tol=2e-4):
    scale = max(float(v.abs().max()) for v in ref_grads.values())
    n = 0
    for name, p in model.gnn_model.named_parameters():
        if not p.numel():
            continue
        ref = ref_grads[name]
        err = float((p.grad.cpu() - ref).abs().max())
        assert err <= tol * float(ref.abs().max()) + 2e-6 * scale, (name, err, float(ref.abs().max()))
        n += 1
    return n


def test_sparse_golden_through_fused_kernels(lba_sparse, protein_params):
    """Reference outputs and reference-autograd gradients vs the whole-pass entry points on their fused plan
    (cgvp_conv_layer_fwd per layer) + the production backward."""
    g = lba_sparse
    N, E = g["x_s"].shape[0], g["edge_index"].shape[1]
    assert ops.fuse_layer(N, E) and ops.VARIANT == "mfma"
    assert _plan(N, E) == (1 + 3 + 2, 1)                     # pass_begin, 3 CSR launches, ONE launch per GVPConvLayer
    model = _protein(protein_params)
    d = _to(dict(x=(T(g["x_s"]), T(g["x_v"])), edge_index=T(g["edge_index"]), ntypes=T(g["ntypes"]),
                 etypes=T(g["etypes"]), eattr=(T(g["e_s"]), T(g["e_v"])), batch=T(g["batch"])))
    with torch.no_grad():                                    # inference launch sequence
        out = model(**d)
    assert rel_err(out, g["out"]) < 2e-5 and rel_err(out, g["out64"]) < 2e-5
    xs, xv = d["x"][0].clone().requires_grad_(), d["x"][1].clone().requires_grad_()
    out = model(**dict(d, x=(xs, xv)))                       # training launch sequence (saves dh), eval-mode dropout
    assert rel_err(out, g["out"]) < 2e-5
    (out * T(g["r"]).to(DEV)).sum().backward()
    assert _check_grads(model, {k[2:]: T(v) for k, v in g.items() if k.startswith("g_")}) >= 60
    assert rel_err(xs.grad, g["gin_x_s"]) < 2e-4 and rel_err(xv.grad, g["gin_x_v"]) < 2e-4


def test_c4_long_graph_full_size(protein_params):
    """BASELINE config 4: one 1,000-residue protein, kNN-20 (20,000 edges), 2 GVP layers -- forward and all
    gradients against the CPU oracle at the configuration's real size."""
    gb = ds.protein_batch(1, 41, length=1000, thresh=20, thresh_type="num")
    assert gb.num_nodes == 1000 and gb.num_edges == 20000
    d = ds.to_torch(gb)
    P = {k: v.clone().requires_grad_(v.numel() > 0) for k, v in protein_params.items()}
    xs, xv = d["x"][0].clone().requires_grad_(), d["x"][1].clone().requires_grad_()
    ref = O.protein_lba_forward(P, (xs, xv), d["edge_index"], d["ntypes"], d["etypes"], d["eattr"])
    r = torch.randn(ref.shape, generator=torch.Generator().manual_seed(2))
    (ref * r).sum().backward()
    model = _protein(protein_params)
    dd = _to(d)
    gxs, gxv = dd["x"][0].clone().requires_grad_(), dd["x"][1].clone().requires_grad_()
    out = model((gxs, gxv), dd["edge_index"], dd["ntypes"], dd["etypes"], eattr=dd["eattr"])
    assert rel_err(out, ref) < 2e-5
    (out * r.to(DEV)).sum().backward()
    _check_grads(model, {k: v.grad for k, v in P.items() if v.numel()})
    assert rel_err(gxs.grad, xs.grad) < 2e-4 and rel_err(gxv.grad, xv.grad) < 2e-4


def _slice_graphs(gb, ids):
    return ds.collate([dict(x_s=gb.x_s[gb.ptr[i]:gb.ptr[i + 1]], x_v=None if gb.x_v is None else gb.x_v[gb.ptr[i]:gb.ptr[i + 1]],
                            edge_index=gb.edge_index[:, gb.eptr[i]:gb.eptr[i + 1]] - gb.ptr[i],
                            e_s=gb.e_s[gb.eptr[i]:gb.eptr[i + 1]],
                            e_v=None if gb.e_v is None else gb.e_v[gb.eptr[i]:gb.eptr[i + 1]],
                            ntypes=gb.ntypes[gb.ptr[i]:gb.ptr[i + 1]], etypes=gb.etypes[gb.eptr[i]:gb.eptr[i + 1]])
                       for i in ids])


def test_c3_kiba_32_pairs_full_size(protein_params, molecule_params):
    """BASELINE config 3, one rank's share: 32 protein/drug pairs with protein lengths drawn from the KIBA
    table (215 ... 4,128 residues; N ~ 23k).  Oracle on a 4-pair subset; the full batch equals its graphs run
    separately and its weight gradients equal the sum over a split (graphs are independent)."""
    lengths = ds.real_lengths("kiba", 32, seed=3)
    pb, mb = ds.pair_batch(32, 17, lengths=lengths)
    assert pb.num_graphs == 32 and 12000 < pb.num_nodes < 40000
    prot, mol = _protein(protein_params), _molecule(molecule_params)
    pparams = [p for p in prot.parameters() if p.numel()]
    mparams = [p for p in mol.parameters() if p.numel()]
    gen = torch.Generator(device=DEV).manual_seed(0)
    r_res = torch.randn(pb.num_nodes, 64, device=DEV, generator=gen)
    r_atm = torch.randn(mb.num_nodes, 64, device=DEV, generator=gen)

    def run(gp, gm, rr, ra):
        dp, dm = _to(ds.to_torch(gp)), _to(ds.to_torch(gm))
        res = prot(dp["x"], dp["edge_index"], dp["ntypes"], dp["etypes"], eattr=dp["eattr"])
        atm = mol(dm["x"], dm["edge_index"], dm["ntypes"], dm["etypes"], eattr=dm["eattr"])
        gr = torch.autograd.grad([res, atm], pparams + mparams, [rr, ra])
        return res.detach(), atm.detach(), gr

    res, atm, g_full = run(pb, mb, r_res, r_atm)
    assert torch.isfinite(res).all() and torch.isfinite(atm).all()
    # oracle on the 4 shortest pairs (keeps the CPU side to seconds)
    ids = [int(i) for i in np.argsort(lengths)[:4]]
    sub_p, sub_m = _slice_graphs(pb, ids), _slice_graphs(mb, ids)
    dp, dm = ds.to_torch(sub_p), ds.to_torch(sub_m)
    ref_res = O.protein_lba_forward(protein_params, dp["x"], dp["edge_index"], dp["ntypes"], dp["etypes"], dp["eattr"])
    ref_atm = O.molecule_gine_forward(molecule_params, dm["x"], dm["edge_index"], dm["ntypes"], dm["etypes"], dm["eattr"])
    got_res = torch.cat([res[int(pb.ptr[i]):int(pb.ptr[i + 1])] for i in ids])
    got_atm = torch.cat([atm[int(mb.ptr[i]):int(mb.ptr[i + 1])] for i in ids])
    assert rel_err(got_res, ref_res) < 2e-5 and rel_err(got_atm, ref_atm) < 2e-5
    # split 16 / 16: outputs concatenate, gradients add
    a, b = list(range(16)), list(range(16, 32))
    na, ma = int(pb.ptr[16]), int(mb.ptr[16])
    res_a, atm_a, g_a = run(_slice_graphs(pb, a), _slice_graphs(mb, a), r_res[:na], r_atm[:ma])
    res_b, atm_b, g_b = run(_slice_graphs(pb, b), _slice_graphs(mb, b), r_res[na:], r_atm[ma:])
    assert rel_err(torch.cat([res_a, res_b]), res) < 1e-6 and rel_err(torch.cat([atm_a, atm_b]), atm) < 1e-6
    scale = max(float(g.abs().max()) for g in g_full)
    for gf, ga, gb_ in zip(g_full, g_a, g_b):
        ref = ga + gb_
        assert float((gf - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-6 * scale


def test_c5_bindingdb_44_bf16_full_size():
    """BASELINE config 5, one rank's share: 32 pairs at BindingDB-scale lengths (mean ~558 residues), CASTER-DTA(4,4)
    (four conv layers; seeded random weights -- no (4,4) checkpoint ships), bf16 variant.  Oracle (same roundings:
    store_dtype + emulate_gemm_dtype, tolerances of tests/test_bf16_storage.py) on the 3 shortest proteins; the full batch
    equals its graphs run separately (rounding-flip level) and its weight gradients equal the sum over a 16 / 16 split."""
    from models.protein_gnn import SelectableProteinModelWrapper
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["protein_gnn_kwargs"]
    kw = dict(kw, num_convs=4)
    for k in ("in_channels", "edge_dim", "hidden_channels", "edge_hidden_channels"):
        kw[k] = tuple(kw[k])
    torch.manual_seed(44)
    prot = SelectableProteinModelWrapper(**kw).to(DEV).eval()
    params = [p for p in prot.parameters() if p.numel()]
    lengths = ds.real_lengths("bindingdb", 32, seed=5)
    pb = ds.protein_batch(32, 23, lengths=lengths)
    assert pb.num_graphs == 32 and 9000 < pb.num_nodes < 40000
    bf = lambda d: {k: (tuple(t.bfloat16() for t in v) if (isinstance(v, tuple) and k in ("x", "eattr")) else v) for k, v in d.items()}
    r_all = torch.randn(pb.num_nodes, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))

    def run(gp, r):
        dp = bf(_to(ds.to_torch(gp)))
        res = prot(dp["x"], dp["edge_index"], dp["ntypes"], dp["etypes"], eattr=dp["eattr"])
        assert res.dtype == torch.bfloat16
        return res.detach().float(), torch.autograd.grad(res, params, r.to(torch.bfloat16))

    res, g_full = run(pb, r_all)
    assert torch.isfinite(res).all() and all(torch.isfinite(g).all() for g in g_full)
    # oracle with the same roundings on the 3 shortest proteins
    ids = [int(i) for i in np.argsort(lengths)[:3]]
    sub = _slice_graphs(pb, ids)
    dc = ds.to_torch(sub)
    rd = lambda t: t.to(torch.bfloat16).float()
    P = {k: v.detach().cpu() for k, v in prot.gnn_model.state_dict().items()}
    with O.emulate_gemm_dtype(torch.bfloat16):
        ref = O.protein_lba_forward(P, tuple(rd(t) for t in dc["x"]), dc["edge_index"], dc["ntypes"], dc["etypes"],
                                    tuple(rd(t) for t in dc["eattr"]), num_convs=4, store_dtype=torch.bfloat16)
    got = torch.cat([res[int(pb.ptr[i]):int(pb.ptr[i + 1])] for i in ids])
    assert rel_err(got, ref) < 1e-2
    # split 16 / 16: outputs concatenate (same per-edge arithmetic), gradients add
    a, b = list(range(16)), list(range(16, 32))
    na = int(pb.ptr[16])
    res_a, g_a = run(_slice_graphs(pb, a), r_all[:na])
    res_b, g_b = run(_slice_graphs(pb, b), r_all[na:])
    assert rel_err(torch.cat([res_a, res_b]), res) < 1e-6
    scale = max(float(g.abs().max()) for g in g_full)
    for gf, ga, gb_ in zip(g_full, g_a, g_b):
        want = ga + gb_
        assert float((gf - want).abs().max()) <= 2e-4 * float(want.abs().max()) + 1e-6 * scale
