"""Varlen residue <-> atom cross attention (csrc/attn_kernels.hip) against stock torch: the op on ragged
batches vs a per-pair dense softmax(QK^T / 4) V, its backward vs torch autograd, the head-averaged weights vs
nn.MultiheadAttention's, and the whole JointGNN (encoders + varlen head) vs the oracle incl. every gradient."""
import json
import os

import numpy as np
import pytest
import torch

import davis_synth as ds
from conftest import GOLDEN, PKG, REPO, rel_err
from gvp_hip import attention_ops  # noqa: F401
from oracle import gvp_oracle as O

DEV = "cuda:0"
H, E = 8, 128


def _ragged(rl, al, seed, dev):
    g = torch.Generator().manual_seed(seed)
    N, Na = sum(rl), sum(al)
    t = [torch.randn(n, E, generator=g) for n in (N, Na, Na, Na, N, N)]
    rptr = torch.tensor([0] + list(np.cumsum(rl)), dtype=torch.long)
    aptr = torch.tensor([0] + list(np.cumsum(al)), dtype=torch.long)
    return [x.to(dev) for x in t], rptr.to(dev), aptr.to(dev)


def _dense_ref(q, k, v, qp, kp):
    """per pair, per head: softmax(q k^T / sqrt(16)) v; also the head-averaged weights per pair"""
    out = torch.zeros_like(q)
    ws = []
    for b in range(len(qp) - 1):
        qs, ks = slice(int(qp[b]), int(qp[b + 1])), slice(int(kp[b]), int(kp[b + 1]))
        qb = q[qs].view(-1, H, 16).transpose(0, 1)
        kb = k[ks].view(-1, H, 16).transpose(0, 1)
        vb = v[ks].view(-1, H, 16).transpose(0, 1)
        p = torch.softmax(qb @ kb.transpose(1, 2) / 4.0, dim=-1)
        out[qs] = (p @ vb).transpose(0, 1).reshape(-1, E)
        ws.append(p.mean(0))
    return out, ws


def test_fake_kernels_cpu():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        f = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device="cuda")
        o_r, o_a, l_r, l_a = torch.ops.caster_gvp.cross_attention(f(70, E), f(9, E), f(9, E), f(9, E), f(70, E), f(70, E),
                                                                  f(3, dt=torch.long), f(3, dt=torch.long), H)
        assert o_r.shape == (70, E) and o_a.shape == (9, E) and l_r.shape == (70, H) and l_a.shape == (9, H)
        w_r, w_a = torch.ops.caster_gvp.cross_attention_weights(f(70, E), f(9, E), l_r, f(9, E), f(70, E), l_a,
                                                                f(3, dt=torch.long), f(3, dt=torch.long), H, 40, 5)
        assert w_r.shape == (2, 40, 5) and w_a.shape == (2, 5, 40)


@pytest.mark.gpu
@pytest.mark.parametrize("rl,al", [([300] * 4, [40, 33, 51, 28]), ([1, 17, 16, 333, 64], [10, 1, 100, 16, 17]),
                                   ([4128, 215], [96, 12])])
def test_cross_attention_forward_backward_weights(rl, al):
    (q_r, k_a, v_a, q_a, k_r, v_r), rptr, aptr = _ragged(rl, al, 5, DEV)
    leaves = [t.clone().requires_grad_() for t in (q_r, k_a, v_a, q_a, k_r, v_r)]
    o_r, o_a, lse_r, lse_a = torch.ops.caster_gvp.cross_attention(*leaves, rptr, aptr, H)
    refl = [t.detach().double().cpu().requires_grad_() for t in leaves]
    ref_r, w_r = _dense_ref(refl[0], refl[1], refl[2], rptr.cpu(), aptr.cpu())
    ref_a, w_a = _dense_ref(refl[3], refl[4], refl[5], aptr.cpu(), rptr.cpu())
    assert rel_err(o_r, ref_r) < 1e-5 and rel_err(o_a, ref_a) < 1e-5
    g = torch.Generator().manual_seed(9)
    g_r, g_a = torch.randn(o_r.shape, generator=g), torch.randn(o_a.shape, generator=g)
    ((o_r * g_r.to(DEV)).sum() + (o_a * g_a.to(DEV)).sum()).backward()
    ((ref_r * g_r.double()).sum() + (ref_a * g_a.double()).sum()).backward()
    for got, want, name in zip(leaves, refl, ("q_r", "k_a", "v_a", "q_a", "k_r", "v_r")):
        assert rel_err(got.grad, want.grad) < 2e-5, name
    rmax, amax = max(rl), max(al)
    d_r, d_a = torch.ops.caster_gvp.cross_attention_weights(q_r, k_a, lse_r.detach(), q_a, k_r, lse_a.detach(), rptr, aptr,
                                                            H, rmax, amax)
    assert d_r.shape == (len(rl), rmax, amax) and d_a.shape == (len(rl), amax, rmax)
    for b, (lr, la) in enumerate(zip(rl, al)):
        assert rel_err(d_r[b, :lr, :la], w_r[b]) < 1e-5 and rel_err(d_a[b, :la, :lr], w_a[b]) < 1e-5
        assert float(d_r[b, lr:].abs().sum()) == 0 and float(d_r[b, :, la:].abs().sum()) == 0
    # run-to-run reproducible (no atomics anywhere)
    again = torch.ops.caster_gvp.cross_attention(*[t.detach() for t in leaves], rptr, aptr, H)
    assert torch.equal(again[0], o_r.detach()) and torch.equal(again[1], o_a.detach())


@pytest.mark.gpu
def test_cross_attention_opcheck():
    (q_r, k_a, v_a, q_a, k_r, v_r), rptr, aptr = _ragged([40, 33], [12, 20], 1, DEV)
    args = tuple(t.clone().requires_grad_() for t in (q_r, k_a, v_a, q_a, k_r, v_r)) + (rptr, aptr, H)
    torch.library.opcheck(torch.ops.caster_gvp.cross_attention.default, args)


@pytest.mark.gpu
def test_module_varlen_equals_dense_mha():
    """CrossAttentionModule.forward_varlen on compact rows == its reference forward (to_dense_batch + stock
    nn.MultiheadAttention with key-padding masks) on the same weights, including the returned weights."""
    from models.joint_gnn import CrossAttentionModule, to_dense_batch
    torch.manual_seed(0)
    mod = CrossAttentionModule(128, 128, 8, 0.0, True, 2, 0.2).to(DEV).eval()
    rl, al = [57, 120, 33], [21, 40, 17]
    g = torch.Generator().manual_seed(3)
    r, m = torch.randn(sum(rl), 128, generator=g).to(DEV), torch.randn(sum(al), 128, generator=g).to(DEV)
    rb = torch.repeat_interleave(torch.arange(3), torch.tensor(rl)).to(DEV)
    mb = torch.repeat_interleave(torch.arange(3), torch.tensor(al)).to(DEV)
    rptr = torch.tensor([0] + list(np.cumsum(rl))).to(DEV)
    aptr = torch.tensor([0] + list(np.cumsum(al))).to(DEV)
    with torch.no_grad():
        e1, e2, (w1, w2) = mod.forward_varlen(r, m, rptr, aptr, need_weights=True)
        dr, rmask = to_dense_batch(r, rb)
        dm, mmask = to_dense_batch(m, mb)
        d1, d2, (x1, x2) = mod(dr, dm, rmask, mmask)
    assert rel_err(e1, d1[rmask]) < 1e-5 and rel_err(e2, d2[mmask]) < 1e-5
    assert w1.shape == x1.shape and w2.shape == x2.shape
    for b, (lr, la) in enumerate(zip(rl, al)):
        assert rel_err(w1[b, :lr, :la], x1[b, :lr, :la]) < 1e-5 and rel_err(w2[b, :la, :lr], x2[b, :la, :lr]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("wgrad_kernel", [False, True])
def test_joint_model_all_gradients_vs_oracle(pretrained, wgrad_kernel, monkeypatch):
    """Encoders (HIP) + varlen head vs the oracle's dense formulation: prediction and the gradient of EVERY one of the
    764,396 parameters on a ragged batch.  wgrad_kernel: the head's per-row Linear layers take their weight / bias
    gradients from caster_gvp::linear_wgrad (what a training batch's row counts select; forced here at 332 rows)."""
    from gvp_hip import head_ops
    from models.joint_gnn import JointGNN
    monkeypatch.setattr(head_ops, "MIN_ROWS", 1 if wgrad_kernel else 1 << 40)
    calls = []                              # layers that took the kernel (through the C++ bridge or the custom op)
    real = head_ops.supported
    monkeypatch.setattr(head_ops, "supported", lambda *a: (lambda ok: (calls.append(1) if ok else None, ok)[1])(real(*a)))
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
    model = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"],
                     **kw["joint_gnn_kwargs"])
    model.load_state_dict(pretrained, strict=True)
    model = model.to(DEV).eval()
    p, m = ds.pair_batch(5, 21, lengths=[40, 75, 33, 120, 64])
    pd, md = ds.to_torch(p), ds.to_torch(m)
    to = lambda d: {k: (tuple(t.to(DEV) for t in v) if isinstance(v, tuple) else v.to(DEV)) for k, v in d.items()}
    dp, dm = to(pd), to(md)
    dp["ptr"], dm["ptr"] = torch.as_tensor(p.ptr).to(DEV), torch.as_tensor(m.ptr).to(DEV)
    y, attn = model(dp, dm)
    assert attn is not None and attn[0][0].shape == (5, 120, int(np.diff(m.ptr).max()))
    P = {k: v.clone().requires_grad_(v.numel() > 0) for k, v in pretrained.items()}
    ref = O.joint_forward(P, pd, md)
    assert rel_err(y, ref) < 1e-4
    r = torch.randn(5, 1, generator=torch.Generator().manual_seed(2))
    (y * r.to(DEV)).sum().backward()
    (ref * r).sum().backward()
    scale = max(float(v.grad.abs().max()) for v in P.values() if v.grad is not None)
    n = 0
    for name, q in model.named_parameters():
        if not q.numel():
            continue
        want = P[name].grad
        err = float((q.grad.cpu() - want).abs().max())
        assert err <= 5e-4 * float(want.abs().max()) + 2e-6 * scale, (name, err, float(want.abs().max()))
        n += 1
    assert n >= 100
    assert (len(calls) >= 12) == wgrad_kernel            # 2 stacks + 2 x (q, kv, out, ff0, ff1)
    model.train()
    model.attention_weights = "auto"
    _, none = model(dp, dm)
    assert none is None                                  # training: the weights are not materialised



@pytest.mark.gpu
@pytest.mark.parametrize("R,I,O", [(19200, 128, 384), (2560, 64, 128), (5003, 256, 128), (1031, 128, 256), (7, 128, 128),
                                   (0, 64, 128), (40000, 128, 128)])
def test_linear_wgrad_kernel_matches_torch(R, I, O):
    """caster_gvp::linear_wgrad (split-row MFMA kernel + fixed-order reduce) vs the matmuls autograd would run, in fp64:
    rows that are no multiple of the 64-row LDS chunk or of the 4-row k-step, one chunk, no rows; run-to-run bitwise."""
    from gvp_hip import head_ops  # noqa: F401
    gen = torch.Generator(device=DEV).manual_seed(R + I + O)
    x = torch.randn(R, I, device=DEV, generator=gen)
    gy = torch.randn(R, O, device=DEV, generator=gen)
    out = torch.ops.caster_gvp.linear_wgrad(x, gy)
    gw, gb = out[:O * I].view(O, I), out[O * I:]
    want_w = (gy.double().t() @ x.double())
    want_b = gy.double().sum(0)
    scale = max(1.0, float(want_w.abs().max()))
    assert float((gw.double() - want_w).abs().max()) <= 2e-6 * scale * max(1, R) ** 0.5
    assert float((gb.double() - want_b).abs().max()) <= 2e-6 * max(1.0, float(want_b.abs().max())) * max(1, R) ** 0.5
    assert torch.equal(out, torch.ops.caster_gvp.linear_wgrad(x, gy))
    torch.library.opcheck(torch.ops.caster_gvp.linear_wgrad.default, (x[:300], gy[:300])) if R >= 300 else None


@pytest.mark.gpu
def test_linear_wgrad_small_input_width_first_then_the_largest_of_its_instantiation():
    """the dynamic-LDS attribute of linear_wgrad_kernel<N> is set once per process and device, so it
    has to cover the largest I the instantiation serves.  In a FRESH process (the attribute is process state): the
    smallest I of each instantiation first, then its largest -- the second launch needs more LDS than the first asked for."""
    import subprocess
    import sys
    code = """
import sys, torch
sys.path[:0] = [%r, %r]
from gvp_hip import head_ops
dev = torch.device('cuda:0')
gen = torch.Generator(device=dev).manual_seed(0)
for I in (16, 80, 96, 144, 160, 256):
    x = torch.randn(2048, I, device=dev, generator=gen); gy = torch.randn(2048, 128, device=dev, generator=gen)
    out = torch.ops.caster_gvp.linear_wgrad(x, gy)
    torch.cuda.synchronize()
    want = gy.double().t() @ x.double()
    err = float((out[:128 * I].view(128, I).double() - want).abs().max())
    assert err <= 2e-6 * float(want.abs().max()) * 2048 ** 0.5, (I, err)
print('ok')
""" % (PKG, REPO)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-3000:]


@pytest.mark.gpu
def test_fast_linear_is_f_linear_with_the_same_gradients():
    from gvp_hip import head_ops
    gen = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(4099, 128, device=DEV, generator=gen, requires_grad=True)
    lin = torch.nn.Linear(128, 256).to(DEV)
    r = torch.randn(4099, 256, device=DEV, generator=gen)
    assert head_ops.supported(x, lin.weight, lin.bias)
    y = head_ops.fast_linear(x, lin.weight, lin.bias)
    gx, gw, gb = torch.autograd.grad(y, [x, lin.weight, lin.bias], r)
    y0 = torch.nn.functional.linear(x, lin.weight, lin.bias)
    hx, hw, hb = torch.autograd.grad(y0, [x, lin.weight, lin.bias], r)
    assert torch.equal(y, y0) and torch.equal(gx, hx)
    assert rel_err(gw, hw) < 1e-5 and rel_err(gb, hb) < 1e-5
    assert not head_ops.supported(x[:100], lin.weight, lin.bias)            # few rows: the library GEMM
    assert not head_ops.supported(x, torch.nn.Linear(128, 100).to(DEV).weight, lin.bias)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,dim", [(19200, 128), (2560, 128), (777, 256), (300, 64), (1025, 512)])
def test_layer_norm_kernels_vs_torch(rows, dim):
    """caster_gvp::layer_norm (csrc/norm_kernels.hip) against F.layer_norm in fp64 -- and no further from it than torch's
    own fp32 kernels are: output, d x, d gamma, d beta; bitwise reproducible (fixed-order reduction)."""
    from gvp_hip import head_ops
    g = torch.Generator(device=DEV).manual_seed(rows + dim)
    x = (torch.randn(rows, dim, device=DEV, generator=g) * 2.0 + 0.5)
    w = torch.randn(dim, device=DEV, generator=g) * 0.3 + 1.0
    b = torch.randn(dim, device=DEV, generator=g) * 0.1
    r = torch.randn(rows, dim, device=DEV, generator=g)
    norm = torch.nn.LayerNorm(dim).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(w)
        norm.bias.copy_(b)

    def run(fn, dtype=torch.float32):
        xx = x.to(dtype).clone().requires_grad_()
        m = norm if dtype == torch.float32 else torch.nn.LayerNorm(dim).to(DEV).double()
        if dtype != torch.float32:
            with torch.no_grad():
                m.weight.copy_(w.double())
                m.bias.copy_(b.double())
        m.zero_grad(set_to_none=True)
        y = fn(xx, m)
        (y * r.to(dtype)).sum().backward()
        return y.detach(), xx.grad, m.weight.grad.clone(), m.bias.grad.clone()

    ref = run(lambda t, m: m(t), torch.float64)
    stock = run(lambda t, m: m(t))
    ours = run(head_ops.fast_layer_norm)
    again = run(head_ops.fast_layer_norm)
    for a, b_ in zip(ours, again):
        assert torch.equal(a, b_)
    for k, (o, s, f) in enumerate(zip(ours, stock, ref)):
        scale = float(f.abs().max())
        e_ours, e_stock = float((o.double() - f).abs().max()), float((s.double() - f).abs().max())
        assert e_ours <= max(2.0 * e_stock, 2e-6 * scale), (k, e_ours, e_stock, scale)


@pytest.mark.gpu
def test_fast_layer_norm_falls_back_outside_its_shapes():
    from gvp_hip import head_ops
    for shape, dim in (((100, 128), 128), ((4000, 96), 96), ((8, 300, 128), 128)):
        norm = torch.nn.LayerNorm(dim).to(DEV)
        x = torch.randn(*shape, device=DEV)
        assert torch.allclose(head_ops.fast_layer_norm(x, norm), norm(x), atol=1e-6)
    norm = torch.nn.LayerNorm(128, elementwise_affine=False).to(DEV)
    x = torch.randn(5000, 128, device=DEV)
    assert torch.allclose(head_ops.fast_layer_norm(x, norm), norm(x), atol=1e-6)
    ident = torch.nn.Identity()
    assert head_ops.fast_layer_norm(x, ident) is x


@pytest.mark.gpu
@pytest.mark.parametrize("rows,dim,p", [(19200, 128, 0.1), (2560, 256, 0.25), (64, 512, 0.5), (333, 8, 0.1)])
def test_fused_dropout_sites_vs_stock_ops_with_the_same_mask(rows, dim, p):
    """caster_gvp::dropout_add / act_dropout (csrc/elementwise_kernels.hip): the factor tensor a site used is read back
    through the op itself (a = 1, x absent), then outputs and gradients are compared with the stock ops under that mask;
    factors are 0 or 1/(1-p) at the right rate, differ between sites and between steps, and the backward regenerates
    exactly the forward's mask."""
    from gvp_hip import head_ops
    dev = torch.device(DEV)
    state = head_ops._head_rng_state(dev)
    pair = torch.ops.caster_gvp.rng_next(state)
    pair2 = torch.ops.caster_gvp.rng_next(state)
    assert int(pair2[1]) == int(pair[1]) + 1 and int(pair2[0]) == int(pair[0])
    none = torch.empty(0, device=DEV)
    ones = torch.ones(rows, dim, device=DEV)
    mask = torch.ops.caster_gvp.dropout_add(ones, none, pair, 3, p)
    vals = torch.unique(mask)
    assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1 / (1 - p)) < 1e-6
    assert abs(float((mask == 0).float().mean()) - p) < max(0.02, 3.0 / (rows * dim) ** 0.5)
    assert not torch.equal(mask, torch.ops.caster_gvp.dropout_add(ones, none, pair, 4, p))       # another site
    assert not torch.equal(mask, torch.ops.caster_gvp.dropout_add(ones, none, pair2, 3, p))      # the next step
    assert torch.equal(mask, torch.ops.caster_gvp.dropout_add(ones, none, pair, 3, p))           # a pure function
    g = torch.Generator(device=DEV).manual_seed(rows)
    a = torch.randn(rows, dim, device=DEV, generator=g, requires_grad=True)
    x = torch.randn(rows, dim, device=DEV, generator=g, requires_grad=True)
    r = torch.randn(rows, dim, device=DEV, generator=g)
    y = torch.ops.caster_gvp.dropout_add(a, x, pair, 3, p)
    ga, gx = torch.autograd.grad((y * r).sum(), [a, x])
    assert torch.allclose(y, x + a * mask, atol=1e-6) and torch.equal(gx, r) and torch.allclose(ga, r * mask, atol=1e-6)
    for slope in (0.0, 0.01):
        t = torch.randn(rows, dim, device=DEV, generator=g, requires_grad=True)
        y = torch.ops.caster_gvp.act_dropout(t, pair, 3, p, slope)
        (gt,) = torch.autograd.grad((y * r).sum(), [t])
        t2 = t.detach().clone().requires_grad_()
        y2 = torch.nn.functional.leaky_relu(t2, slope) * mask
        (gt2,) = torch.autograd.grad((y2 * r).sum(), [t2])
        assert torch.allclose(y, y2, atol=1e-6) and torch.allclose(gt, gt2, atol=1e-6)
    # p = 0: no generator needed, identity factors
    assert torch.equal(torch.ops.caster_gvp.dropout_add(a.detach(), x.detach(), pair, 1, 0.0), x.detach() + a.detach())


@pytest.mark.gpu
def test_bridge_head_ops_equal_the_custom_ops():
    """Eager mode drives the head's row-wise ops and the attention core through C++ autograd functions of the bridge
    (csrc/torch_bridge.cpp), torch.compile through the torch.library ops: same C entry points, so outputs and every
    gradient must agree bitwise."""
    from gvp_hip import _lib, attention_ops, head_ops  # noqa: F401
    br = _lib.bridge()
    if br is None or not hasattr(br, "head_cross_attention"):
        pytest.skip("C++ bridge not built")
    g = torch.Generator(device=DEV).manual_seed(3)
    f = lambda *s: torch.randn(*s, device=DEV, generator=g)

    def both(run_a, run_b, leaves):
        outs = []
        for run in (run_a, run_b):
            ls = [t.clone().requires_grad_() for t in leaves]
            y = run(*ls)
            ys = y if isinstance(y, (tuple, list)) else (y,)
            ys = [t for t in ys if t.requires_grad]
            torch.autograd.backward(ys, [torch.ones_like(t) * 0.5 for t in ys])
            outs.append(([t.detach() for t in ys], [t.grad for t in ls]))
        for a, b in zip(outs[0][0] + outs[0][1], outs[1][0] + outs[1][1]):
            assert torch.equal(a, b)

    x, w, b = f(2048, 128), f(256, 128), f(256)
    both(lambda x, w, b: br.head_linear(x, w, b), lambda x, w, b: head_ops._FastLinear.apply(x, w, b), [x, w, b])
    gam, bet = f(128), f(128)
    both(lambda x, w, b: br.head_layer_norm(x, w, b, 1e-5), lambda x, w, b: torch.ops.caster_gvp.layer_norm(x, w, b, 1e-5)[0],
         [x, gam, bet])
    state = head_ops._head_rng_state(torch.device(DEV))
    pair = br.head_rng_next(state)
    a = f(2048, 128)
    both(lambda a, x: br.head_dropout_add(a, x, pair, 3, 0.2), lambda a, x: torch.ops.caster_gvp.dropout_add(a, x, pair, 3, 0.2),
         [a, x])
    both(lambda t: br.head_act_dropout(t, pair, 5, 0.2, 0.01), lambda t: torch.ops.caster_gvp.act_dropout(t, pair, 5, 0.2, 0.01),
         [a])
    (q_r, k_a, v_a, q_a, k_r, v_r), rptr, aptr = _ragged([40, 17, 300], [9, 33, 16], 7, DEV)
    both(lambda *t: br.head_cross_attention(*t, rptr, aptr, H)[:2], lambda *t: torch.ops.caster_gvp.cross_attention(*t, rptr, aptr, H)[:2],
         [q_r, k_a, v_a, q_a, k_r, v_r])
