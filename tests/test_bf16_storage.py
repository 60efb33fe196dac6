"""bf16 storage + bf16 matrix-core operands, fp32 accumulate (BASELINE config 5; cgvp_dims.storage = CGVP_BF16):
features, node rows, the edge-embedding store and the residue embeddings live in HBM as bfloat16; every channel GEMM with
more than 4 input channels runs on v_mfma_f32_16x16x16_bf16 (operands rounded to bf16 -- activations, weights and, in
the backward, gradients -- products accumulated in fp32); LayerNorm statistics, norms, gates, biases, the one-hot type
columns and every gradient BUFFER stay fp32.

Checker: the fp32 oracle on the SAME bf16-rounded inputs, rounding the same tensors at the same points --
`store_dtype=torch.bfloat16` (stage hand-offs: node embedding, edge embedding, aggregated messages, layer outputs,
result; straight-through gradient) inside `emulate_gemm_dtype(torch.bfloat16)` (every GVP Linear: input, weight and
incoming gradient rounded, <= 4-input-channel Linears exact, type columns exact).  What is left between the kernels and
this emulation is fp32 summation order plus the occasional 1-ulp bf16 flip at a rounding boundary (2^-8 relative on
single elements, and a ReLU whose input sits within that of zero): measured on MI355X 3.4e-3 of the output scale and
<= 1e-2 in the L2 norm of every weight gradient (davis_2_2).  Stated tolerance: outputs 1e-2, every gradient tensor
2e-2 in L2 (+ the absolute floor for analytically-zero gradients).  For scale: the SAME comparison against the
unrounded fp32 oracle gives 4e-3 (outputs) and 6-30 % (gradients of a random upstream gradient: ReLU masks flip) --
that is the price of bf16 operands, not of this implementation; the fp32-storage kernels are the exact path.
Against the fp32-storage kernels on the same inputs the outputs stay within 3e-2; the bf16 path repeats bitwise."""
import json
import os

import numpy as np
import pytest
import torch

import davis_synth as ds
from conftest import GOLDEN, rel_err
from oracle import gvp_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FWD_TOL, GRAD_TOL, VS_F32 = 1e-2, 2e-2, 3e-2


def _encoder(state=None, num_convs=2, seed=None):
    from models.protein_gnn import SelectableProteinModelWrapper
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["protein_gnn_kwargs"]
    kw = dict(kw, num_convs=num_convs)
    for k in ("in_channels", "edge_dim", "hidden_channels", "edge_hidden_channels"):
        kw[k] = tuple(kw[k])
    if seed is not None:
        torch.manual_seed(seed)
    m = SelectableProteinModelWrapper(**kw)
    if state is not None:
        m.load_state_dict({"gnn_model." + k: v for k, v in state.items()})
    return m.to(DEV).eval()


def _bf(d, dev=None):
    """features rounded to bf16 (kept as bf16 on `dev`, or widened back to fp32 on the CPU for the oracle)"""
    c = lambda t: t.to(torch.bfloat16).to(dev) if dev else t.to(torch.bfloat16).float()
    return {k: (tuple(c(t) for t in v) if isinstance(v, tuple) else (v.to(dev) if dev else v)) for k, v in d.items()}


@pytest.mark.parametrize("case", ["davis_2_2", "knn_unfused", "depth4_ragged"])
def test_bf16_forward_backward_vs_oracle(protein_params, case):
    if case == "davis_2_2":
        gb, nc, state = ds.protein_batch(8, 3), 2, protein_params
    elif case == "knn_unfused":
        gb, nc, state = ds.protein_batch(2, 5, length=120, thresh=12, thresh_type="num"), 2, protein_params
    else:
        gb, nc, state = ds.protein_batch(4, 7, lengths=[1, 40, 77, 33], thresh=6.0), 4, None      # CASTER-DTA(4,4) depth
    model = _encoder(state, nc, seed=11)
    d = ds.to_torch(gb)
    dd, dc = _bf(d, DEV), _bf(d)
    xs, xv = dd["x"][0].clone().requires_grad_(), dd["x"][1].clone().requires_grad_()
    out = model((xs, xv), dd["edge_index"], dd["ntypes"], dd["etypes"], eattr=dd["eattr"])
    assert out.dtype == torch.bfloat16 and out.shape == (gb.num_nodes, 64)
    P = {k: v.detach().cpu().clone().requires_grad_(v.numel() > 0) for k, v in model.gnn_model.state_dict().items()}
    rxs, rxv = dc["x"][0].clone().requires_grad_(), dc["x"][1].clone().requires_grad_()
    with O.emulate_gemm_dtype(torch.bfloat16):            # bf16 matrix-core operands, fp32 accumulate (see the oracle)
        ref = O.protein_lba_forward(P, (rxs, rxv), dc["edge_index"], dc["ntypes"], dc["etypes"], dc["eattr"], num_convs=nc,
                                    store_dtype=torch.bfloat16)
        assert rel_err(out.float(), ref) < FWD_TOL
        r = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3))
        (out.float() * r.to(DEV)).sum().backward()
        (ref * r).sum().backward()
    scale = max(float(v.grad.abs().max()) for v in P.values() if v.grad is not None)
    worst = {}
    for name, p in model.gnn_model.named_parameters():
        if p.numel():
            want = P[name].grad
            assert p.grad.dtype == torch.float32                      # weight gradients are fp32
            worst[name] = (float((p.grad.cpu() - want).norm()) - 2e-3 * scale) / max(float(want.norm()), 1e-30)
    bad = {k: round(v, 4) for k, v in worst.items() if v > GRAD_TOL}
    assert not bad, (bad, sorted(round(v, 4) for v in worst.values())[-8:])
    assert xs.grad.dtype == torch.bfloat16
    l2 = lambda a, b: float((a.cpu().float() - b).norm() / b.norm())
    assert l2(xs.grad, rxs.grad) < GRAD_TOL and l2(xv.grad, rxv.grad) < GRAD_TOL
    # same kernels in fp32 storage on the same (bf16-representable) inputs: agree to the same bound; bf16 run repeats bitwise
    f32 = {k: (tuple(t.float() for t in v) if isinstance(v, tuple) else v) for k, v in dd.items()}
    with torch.no_grad():
        o32 = model(f32["x"], f32["edge_index"], f32["ntypes"], f32["etypes"], eattr=f32["eattr"])
        o16 = model(dd["x"], dd["edge_index"], dd["ntypes"], dd["etypes"], eattr=dd["eattr"])
        again = model(dd["x"], dd["edge_index"], dd["ntypes"], dd["etypes"], eattr=dd["eattr"])
    assert o32.dtype == torch.float32 and rel_err(o16.float(), o32) < VS_F32
    assert torch.equal(o16, again)


def test_bf16_training_mode_and_joint_head(pretrained):
    """bf16-storage encoders inside JointGNN (fp32 head): a training step with in-kernel dropout runs and every
    parameter receives a finite gradient."""
    from models.joint_gnn import JointGNN
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
    model = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"],
                     **kw["joint_gnn_kwargs"])
    model.load_state_dict(pretrained, strict=True)
    model = model.to(DEV).train()
    p, m = ds.pair_batch(4, 5, lengths=[40, 55, 33, 70])
    pd = _bf(ds.to_torch(p), DEV)
    md = {k: v.to(DEV) for k, v in ds.to_torch(m).items()}
    pred, _ = model(pd, md)
    assert pred.dtype == torch.float32 and pred.shape == (4, 1)
    pred.square().mean().backward()
    bad = [n for n, q in model.named_parameters() if q.numel() and (q.grad is None or not torch.isfinite(q.grad).all())]
    assert not bad, bad


def test_bf16_kernels_vs_reference_amp_fixture(protein_params, lba_sparse):
    """g1 against the REFERENCE's own reduced-precision numerics: tests/golden/lba_amp_bf16.npz holds the unmodified
    reference encoder on the lba_sparse batch under torch.autocast("cpu", dtype=torch.bfloat16) -- what
    train_model.py:561 runs on a CPU device (output bf16, reference-autograd gradients).  The reference's autocast
    rounds every nn.Linear output to bf16 and keeps everything between them in whatever dtype falls out; the kernels
    round at different points (bf16 stage hand-offs, bf16 MFMA operands, fp32 accumulate / norms / gates), so the two
    are not bit-comparable; what can be stated is that the kernels sit as close to the fp32 reference as the
    reference's own AMP does, and within the sum of both distances of the AMP result.
    Measured on MI355X (printed below): AMP-ref vs fp32-ref 7.3e-3 (fixture); kernels vs fp32-ref 5.5e-3; kernels vs
    AMP-ref 7.4e-3 -- asserted at the bounds KERNEL_VS_F32 / KERNEL_VS_AMP; weight gradients (L2 per tensor, relative
    to the fp32 reference gradient): kernels median 6.1e-2, reference AMP median 5.0e-2."""
    from conftest import load_npz
    amp, g = load_npz("lba_amp_bf16.npz"), lba_sparse
    assert str(amp["out_dtype"]) == "torch.bfloat16"
    T = torch.from_numpy
    ref32, ref_amp = T(g["out"]), T(amp["out"])
    amp_vs_f32 = rel_err(ref_amp, ref32)
    assert 1e-3 < amp_vs_f32 < 2e-2
    model = _encoder(protein_params)
    b = lambda a: T(a).to(DEV).to(torch.bfloat16)
    xs, xv = b(g["x_s"]).requires_grad_(), b(g["x_v"]).requires_grad_()
    out = model((xs, xv), T(g["edge_index"]).to(DEV), T(g["ntypes"]).to(DEV), T(g["etypes"]).to(DEV),
                eattr=(b(g["e_s"]), b(g["e_v"])))
    assert out.dtype == torch.bfloat16
    k_vs_f32, k_vs_amp = rel_err(out.float(), ref32), rel_err(out.float(), ref_amp)
    print(f"\nbf16 kernels vs fp32 reference {k_vs_f32:.2e}; reference AMP vs fp32 reference {amp_vs_f32:.2e}; "
          f"kernels vs reference AMP {k_vs_amp:.2e}")
    KERNEL_VS_F32, KERNEL_VS_AMP = 1.5e-2, 2.5e-2
    assert k_vs_f32 < KERNEL_VS_F32 and k_vs_amp < KERNEL_VS_AMP
    assert k_vs_f32 < 2.0 * amp_vs_f32                         # no further from the exact result than the reference's AMP (x2 slack)
    # gradients of the fixture's upstream gradient r (same r as the AMP fixture): L2 distance per weight tensor,
    # relative to the tensor's own norm, kernels vs fp32 reference next to reference-AMP vs fp32 reference
    (out.float() * T(amp["r"]).to(DEV)).sum().backward()
    rows = []
    scale = max(float(np.abs(v).max()) for k, v in g.items() if k.startswith("g_"))
    for name, p in model.gnn_model.named_parameters():
        if not p.numel():
            continue
        g32, gamp, gk = T(g["g_" + name]).double(), T(amp["g_" + name]).double(), p.grad.cpu().double()
        n = float(g32.norm()) + 1e-3 * scale * p.numel() ** 0.5          # floor: analytically-zero gradients are rounding noise
        rows.append((name, float((gk - g32).norm()) / n, float((gamp - g32).norm()) / n))
    worst_k, worst_a = max(r[1] for r in rows), max(r[2] for r in rows)
    med_k, med_a = float(np.median([r[1] for r in rows])), float(np.median([r[2] for r in rows]))
    print(f"weight gradients, L2 relative to the fp32 reference gradients: kernels median {med_k:.2e} worst {worst_k:.2e}; "
          f"reference AMP median {med_a:.2e} worst {worst_a:.2e}")
    assert med_k < 2.0 * med_a + 1e-2 and worst_k < 2.0 * worst_a + 5e-2
