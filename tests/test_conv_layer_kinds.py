"""Stand-alone GVPConv / GVPConvLayer on the tile kernels, for the three layer kinds of the reference
(gvp_layers.py:340-366: (relu, None)+gate -- CASTER-DTA; (relu, sigmoid) -- the defaults, CPD-style stacks;
(None, None) -- PocketMiner-style stacks), survey rows a7 / a11 / f-4.

Checker: this package's own tensor-op composition of the SAME module on the SAME device (`conv_layer_ops.ENABLED =
False`), which tests/test_gvp_stacks.py and tests/test_gvp_units pin against the reference's classes on the CPU.
Outputs to 2e-5, gradients of every parameter, of the node features and of the edge embedding to 2e-4 of the largest
gradient of the tensor."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

KINDS = {
    "gated": dict(activations=(F.relu, None), vector_gate=True),
    "gvpdef": dict(),                                            # (relu, sigmoid), no gate: the constructor defaults
    "linear": dict(activations=(None, None)),
}


def _graph(n, e, seed, isolated=7):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.randint(0, n - isolated, (e,), generator=g)      # the last `isolated` nodes receive nothing
    return torch.stack([src, dst]).to(DEV)


def _feats(n, dims, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (scale * torch.randn(n, dims[0], generator=g)).to(DEV), (scale * torch.randn(n, dims[1], 3, generator=g)).to(DEV)


def _layer(kind, se, aggr=None, autoregressive=False, drop_rate=0.0, seed=0):
    import models.gvp_layers as gvp
    torch.manual_seed(seed)
    layer = gvp.GVPConvLayer((16, 4), (se, 1), drop_rate=drop_rate, aggr=aggr, autoregressive=autoregressive, **KINDS[kind])
    with torch.no_grad():
        for k, p in layer.named_parameters():                   # default init: identity LayerNorm, zero-mean small weights
            if "scalar_norm" in k:
                p.add_(0.2 * torch.randn_like(p))
            elif p.numel():
                p.mul_(1.5)
    return layer.to(DEV)


def _run(layer, x, ei, e, r, enabled, **kw):
    from gvp_hip import conv_layer_ops as K
    leaves = [t.detach().clone().requires_grad_(True) for t in (*x, *e)]
    extra = {}
    if "autoregressive_x" in kw:
        ar = [t.detach().clone().requires_grad_(True) for t in kw["autoregressive_x"]]
        leaves += ar
        extra["autoregressive_x"] = tuple(ar)
    layer.zero_grad(set_to_none=True)
    old = K.ENABLED
    K.ENABLED = enabled
    try:
        out = layer((leaves[0], leaves[1]), ei, (leaves[2], leaves[3]), **extra)
        ((out[0] * r[0]).sum() + (out[1] * r[1]).sum()).backward()
    finally:
        K.ENABLED = old
    grads = {k: p.grad.detach().clone() for k, p in layer.named_parameters() if p.numel()}
    grads.update({f"input{i}": t.grad.detach().clone() for i, t in enumerate(leaves)})
    return (out[0].detach(), out[1].detach()), grads


def _compare(a, b, tol_out=2e-5, tol_grad=2e-4):
    (so, vo), ga = a
    (sr, vr), gb = b
    for x, y in ((so, sr), (vo, vr)):
        assert float((x - y).abs().max()) <= tol_out * float(y.abs().max()) + 1e-6
    assert set(ga) == set(gb)
    scale = max(float(v.abs().max()) for v in gb.values())
    for k in gb:
        err = float((ga[k] - gb[k]).abs().max())
        assert err <= tol_grad * float(gb[k].abs().max()) + 2e-6 * scale, (k, err, float(gb[k].abs().max()))


@pytest.mark.parametrize("kind", list(KINDS))
@pytest.mark.parametrize("se,aggr", [(32, None), (20, "add")])
def test_layer_kinds_match_the_composition(kind, se, aggr):
    from gvp_hip import conv_layer_ops as K
    layer = _layer(kind, se, aggr=aggr).eval()
    assert K.conv_kind(layer.conv) == K.node_kind(layer) == {"gated": 0, "gvpdef": 1, "linear": 2}[kind]
    n, e = 333, 1900
    ei = _graph(n, e, 3)
    x, ea, r = _feats(n, (16, 4), 4), _feats(e, (se, 1), 5), _feats(n, (16, 4), 6)
    _compare(_run(layer, x, ei, ea, r, True), _run(layer, x, ei, ea, r, False))


def test_non_gated_kinds_leave_no_gradient_in_the_gate_slots():
    """The arena keeps wsv slots for every kind; for the un-gated kinds the kernels must neither read them nor write a
    gradient there: poison them in a packed arena and compare with the clean one."""
    from gvp_hip import conv_layer_ops as K, ops
    layer = _layer("gvpdef", 32).eval()
    n, e = 200, 900
    ei = _graph(n, e, 8)
    x, ea = _feats(n, (16, 4), 9), _feats(e, (32, 1), 10)
    kind = K.conv_kind(layer.conv)
    clean = K.pack_arena(kind, conv=layer.conv, layer=layer).detach()
    zero_slots = clean == 0
    dirty = torch.where(zero_slots, torch.full_like(clean, 3.0), clean).requires_grad_(True)
    clean.requires_grad_(True)
    outs = []
    for arena in (clean, dirty):
        image = K.prepare(kind, arena)
        csr = ops.build_csr(ei, n)
        dh = K._ConvFn.apply(arena, K.rows_from_tuple(x), K.edge_rows(ea, csr.eperm, e), image, csr.rowptr, csr.esrc,
                             csr.edst, kind, True)
        out = K._NodeFn.apply(arena, K.rows_from_tuple(x), dh, image, None, None, kind)
        out.square().sum().backward()
        outs.append(out.detach())
    assert torch.equal(outs[0], outs[1])
    # every slot that is not a parameter of the layer (embed / head blocks, wsv of the un-gated GVPs) has zero gradient
    assert float(clean.grad[zero_slots].abs().max()) == 0.0 and float(dirty.grad[zero_slots].abs().max()) == 0.0
    assert torch.equal(clean.grad, dirty.grad) and float(clean.grad.abs().max()) > 0


def test_autoregressive_layer_runs_two_masked_passes_on_the_kernels(monkeypatch):
    from gvp_hip import conv_layer_ops as K
    layer = _layer("gvpdef", 32, autoregressive=True).eval()
    n, e = 257, 1500
    ei = _graph(n, e, 11, isolated=3)
    x, ea, r, ar = _feats(n, (16, 4), 12), _feats(e, (32, 1), 13), _feats(n, (16, 4), 14), _feats(n, (16, 4), 15)
    calls = []
    conv_message = K.conv_message
    monkeypatch.setattr(K, "conv_message", lambda *a, **k: (calls.append(int(a[3].shape[1])), conv_message(*a, **k))[1])
    got = _run(layer, x, ei, ea, r, True, autoregressive_x=ar)
    fwd = int((ei[0] < ei[1]).sum())
    assert calls == [fwd, e - fwd] and 0 < fwd < e
    _compare(got, _run(layer, x, ei, ea, r, False, autoregressive_x=ar))


@pytest.mark.parametrize("kind", ["gvpdef", "linear"])
def test_training_mode_dropout_masks(kind):
    """Explicit masks: the kernels' node update against the composition with the same masks applied by hand; drawn
    masks: factors are 0 or 1/(1-p), vector channels dropped whole, about p of them."""
    from gvp_hip import conv_layer_ops as K
    import models.gvp_layers as gvp
    p = 0.25
    layer = _layer(kind, 32, drop_rate=p).train()
    n, e = 300, 1400
    ei = _graph(n, e, 21)
    x, ea, r = _feats(n, (16, 4), 22), _feats(e, (32, 1), 23), _feats(n, (16, 4), 24)
    m0, m1 = K.draw_masks(n, p, DEV), K.draw_masks(n, p, DEV)
    for m in (m0, m1):
        vals = np.unique(m.cpu().numpy())
        assert len(vals) == 2 and vals[0] == 0.0 and abs(float(vals[1]) - 1 / (1 - p)) < 1e-6
        assert abs(float((m == 0).float().mean()) - p) < 0.03
    assert not torch.equal(m0, m1)

    def run(kernels):
        leaves = [t.detach().clone().requires_grad_(True) for t in x]
        layer.zero_grad(set_to_none=True)
        K.ENABLED = kernels
        try:
            dh = layer.conv((leaves[0], leaves[1]), ei, ea)
            if kernels:
                out = K.node_update(layer, K.node_kind(layer), (leaves[0], leaves[1]), K.rows_from_tuple(dh), masks=(m0, m1))
            else:
                def drop(t, m):
                    return t[0] * m[:, :16], t[1] * m[:, 16:, None]
                h1 = layer.norm[0](gvp.tuple_sum((leaves[0], leaves[1]), drop(dh, m0)))
                out = layer.norm[1](gvp.tuple_sum(h1, drop(layer.ff_func(h1), m1)))
            ((out[0] * r[0]).sum() + (out[1] * r[1]).sum()).backward()
        finally:
            K.ENABLED = True
        grads = {k: q.grad.detach().clone() for k, q in layer.named_parameters() if q.numel()}
        grads.update({f"input{i}": t.grad.detach().clone() for i, t in enumerate(leaves)})
        return (out[0].detach(), out[1].detach()), grads

    _compare(run(True), run(False))
    # the module's own training-mode forward draws fresh masks: differs from eval, finite, right shape
    out_t = layer(x, ei, ea)
    out_e = layer.eval()(x, ei, ea)
    assert torch.isfinite(out_t[0]).all() and out_t[1].shape == (n, 4, 3) and not torch.allclose(out_t[0], out_e[0])


def test_node_mask_updates_only_the_selected_rows_on_the_kernels(monkeypatch):
    """`node_mask` (gvp_layers.py:403-414): messages come from every node, only the masked nodes are updated."""
    from gvp_hip import conv_layer_ops as K
    layer = _layer("gvpdef", 32).eval()
    n, e = 210, 1000
    ei = _graph(n, e, 51)
    x, ea = _feats(n, (16, 4), 52), _feats(e, (32, 1), 53)
    mask = (torch.arange(n, device=DEV) % 3) != 1
    sizes = []
    node_update = K.node_update
    monkeypatch.setattr(K, "node_update", lambda *a, **k: (sizes.append(int(a[2][0].shape[0])), node_update(*a, **k))[1])
    outs = []
    for enabled in (True, False):
        K.ENABLED = enabled
        try:
            xin = (x[0].clone(), x[1].clone())
            with torch.no_grad():
                outs.append(layer(xin, ei, ea, node_mask=mask))
        finally:
            K.ENABLED = True
    assert sizes == [int(mask.sum())]
    for a, b, orig in zip(outs[0], outs[1], x):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
        assert torch.equal(a[~mask], orig[~mask]) and not torch.allclose(a[mask], orig[mask])


def test_cases_outside_the_compiled_set_use_the_composition(monkeypatch):
    """Other widths / activation mixes are not silently mis-computed: they never reach the kernels."""
    import models.gvp_layers as gvp
    from gvp_hip import conv_layer_ops as K

    def boom(*a, **k):
        raise AssertionError("kernel path taken")
    monkeypatch.setattr(K, "conv_message", boom)
    monkeypatch.setattr(K, "node_update", boom)
    n, e = 64, 300
    ei = _graph(n, e, 31, isolated=0)
    torch.manual_seed(0)
    wide = gvp.GVPConvLayer((32, 4), (32, 1)).to(DEV).eval()                      # 32 hidden scalars
    out = wide(_feats(n, (32, 4), 1), ei, _feats(e, (32, 1), 2))
    assert out[0].shape == (n, 32)
    mixed = gvp.GVPConvLayer((16, 4), (32, 1), activations=(F.relu, torch.sigmoid), vector_gate=True).to(DEV).eval()
    assert K.conv_kind(mixed.conv) is None and K.node_kind(mixed) is None         # gate fed through sigmoid: no such kernel
    assert mixed(_feats(n, (16, 4), 3), ei, _feats(e, (32, 1), 4))[0].shape == (n, 16)
    wide_vec = gvp.GVPConv((16, 4), (16, 4), (32, 2)).to(DEV).eval()              # two edge vector channels
    assert K.conv_kind(wide_vec) is None
    assert wide_vec(_feats(n, (16, 4), 5), ei, _feats(e, (32, 2), 6))[0].shape == (n, 16)


@pytest.mark.parametrize("se", [52, 40, 64])
@pytest.mark.parametrize("kind", ["gated", "gvpdef", "linear"])
def test_more_than_32_edge_scalars_run_folded_on_the_kernels(kind, se, monkeypatch):
    """Round 4 (survey f-4): an edge embedding wider than the compiled 32 scalars -- the default CPD decoder's 32 + 20
    (protein_gnn.py:566-570) -- runs on the conv kernels with the edge scalars folded through message_func.0's `ws`
    (gvp_hip/conv_layer_ops.py, "FOLDED").  Messages and every gradient (weights, node features, edge features) against
    the tensor-op composition of the same module."""
    import models.gvp_layers as gvp
    from gvp_hip import conv_layer_ops as K
    acts = {"gated": dict(activations=(F.relu, None), vector_gate=True), "gvpdef": dict(activations=(F.relu, torch.sigmoid)),
            "linear": dict(activations=(None, None))}[kind]
    torch.manual_seed(7)
    conv = gvp.GVPConv((16, 4), (16, 4), (se, 1), aggr="mean", **acts).to(DEV)
    assert K.conv_kind(conv) is not None and K.folds_edges(conv)
    n, e = 150, 1100
    ei = _graph(n, e, 9, isolated=3)
    calls = []
    conv_message = K.conv_message
    monkeypatch.setattr(K, "conv_message", lambda *a, **k: (calls.append(1), conv_message(*a, **k))[1])
    res = {}
    for tag, on in (("kernels", True), ("composition", False)):
        monkeypatch.setattr(K, "ENABLED", on)
        x = tuple(t.clone().requires_grad_() for t in _feats(n, (16, 4), 1))
        ea = tuple(t.clone().requires_grad_() for t in _feats(e, (se, 1), 2))
        conv.zero_grad(set_to_none=True)
        out = conv(x, ei, ea)
        r = _feats(n, (16, 4), 3)
        ((out[0] * r[0]).sum() + (out[1] * r[1]).sum()).backward()
        res[tag] = ([out[0].detach(), out[1].detach(), x[0].grad, x[1].grad, ea[0].grad, ea[1].grad]
                    + [p.grad.clone() for p in conv.parameters() if p.numel()])
    assert len(calls) == 1                                       # the kernel path ran once, the composition never reached it
    scale = max(float(t.abs().max()) for t in res["composition"][6:])
    for a, b in zip(res["kernels"], res["composition"]):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 2e-6 * scale


def test_layer_kind_is_rejected_where_the_library_has_no_such_kernel():
    """cgvp_dims.layer_kind != GATED: whole passes, embeddings and the fused layer return UNSUPPORTED_DIMS; bf16 storage
    exists for the gated kind only."""
    import ctypes as C
    from gvp_hip import _lib, ops
    L = _lib.lib()
    dims = ops.make_dims(layer_kind=ops.LAYER_GVPDEF)
    layout = ops.lba_layout(dims, 0, 0, 1)                        # layout / image queries accept every kind
    assert layout.total == ops.lba_layout(ops.make_dims(), 0, 0, 1).total
    ws = _lib.LbaFwdWs()
    assert L.cgvp_lba_fwd_workspace(C.byref(dims), C.byref(layout), 10, 20, 1, C.byref(ws)) == -2
    assert L.cgvp_lba_bwd_workspace_bytes(C.byref(dims), C.byref(layout), 10, 20) == -2
    bad = ops.make_dims(layer_kind=ops.LAYER_LINEAR, storage=ops.BF16)
    out = _lib.Layout()
    assert L.cgvp_lba_layout(C.byref(bad), 0, 0, 1, C.byref(out)) == -2
    assert L.cgvp_lba_layout(C.byref(ops.make_dims(layer_kind=3)), 0, 0, 1, C.byref(out)) == -2


def test_empty_graph_and_isolated_nodes():
    layer = _layer("linear", 32).eval()
    n = 40
    x = _feats(n, (16, 4), 41)
    ei = torch.zeros(2, 0, dtype=torch.long, device=DEV)
    out = layer(x, ei, _feats(0, (32, 1), 42))
    from gvp_hip import conv_layer_ops as K
    K.ENABLED = False
    try:
        ref = layer(x, ei, _feats(0, (32, 1), 42))
    finally:
        K.ENABLED = True
    assert float((out[0] - ref[0]).detach().abs().max()) < 1e-5 and float((out[1] - ref[1]).detach().abs().max()) < 1e-5


@pytest.mark.parametrize("kind", list(KINDS))
def test_rotation_equivariance_at_full_batch_size(kind):
    """Size-independent property at the headline batch size (19,200 nodes, ~57k edges): scalar outputs are invariant and
    vector outputs co-rotate when every input vector is rotated -- for each layer kind on the kernels, forward and the
    gradient w.r.t. the node vectors (which must co-rotate too)."""
    import davis_synth as ds
    layer = _layer(kind, 32).eval()
    pb = ds.protein_batch(64, 3)
    ei = torch.from_numpy(pb.edge_index).to(DEV)
    n, e = pb.num_nodes, ei.shape[1]
    assert n == 19200 and e > 50000
    x, ea, r = _feats(n, (16, 4), 61), _feats(e, (32, 1), 62), _feats(n, (16, 4), 63)
    g = torch.Generator().manual_seed(7)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    R = q.to(DEV)
    rot = lambda t: (t[0], t[1] @ R.T)

    def run(xin, ein, rin):
        xs, xv = xin[0].clone().requires_grad_(), xin[1].clone().requires_grad_()
        out = layer((xs, xv), ei, ein)
        ((out[0] * rin[0]).sum() + (out[1] * rin[1]).sum()).backward()
        return out[0].detach(), out[1].detach(), xs.grad, xv.grad

    s0, v0, gs0, gv0 = run(x, ea, r)
    s1, v1, gs1, gv1 = run(rot(x), rot(ea), rot(r))
    tol = lambda ref: 5e-5 * float(ref.abs().max())
    assert float((s1 - s0).abs().max()) <= tol(s0)
    assert float((v1 - v0 @ R.T).abs().max()) <= tol(v0)
    assert float((gs1 - gs0).abs().max()) <= tol(gs0) * 4
    assert float((gv1 - gv0 @ R.T).abs().max()) <= tol(gv0) * 4
    assert float(v0.abs().max()) > 0 and not torch.allclose(v1, v0, atol=1e-3)
