"""CPU oracle for the CASTER-DTA encoder hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``caster-dta_amd/``) never imports anything from
``oracle/`` and raises when its HIP library is missing.

It restates, op for op and in the reference's own operation order (index_select
gathers -> cat -> Linear -> index_add scatter), the algorithm of

    /root/reference/models/gvp_layers.py     (GVP, LayerNorm, GVPConv, GVPConvLayer)
    /root/reference/models/protein_gnn.py    (VectorProteinGNN_LBAModel)
    /root/reference/models/molecule_gnn.py   (HomoMoleculeGNN_GINE)
    /root/reference/models/joint_gnn.py      (JointGNN head, stock torch.nn)

as plain functions over a flat ``{state_dict key: tensor}`` mapping, so the same
weights drive the oracle, the HIP path and the reference.  Works in fp32 or
fp64 (dtype follows the inputs) and is differentiable through torch autograd,
which makes it the gradient oracle for the HIP backward kernels as well.

Pinning status
--------------
* Protein path (GVP / LayerNorm / GVPConvLayer / LBA model): PINNED.  Checked by
  ``tests/test_oracle_golden.py`` against ``tests/golden/*.npz`` -- outputs,
  per-stage intermediates and autograd gradients produced by the unmodified
  reference modules run in the build container with the pretrained checkpoint
  (generator: ``tests/golden/make_golden.py``).
* ``MessagePassing.propagate`` semantics (gather ``x[edge_index[0]]`` as ``_j``,
  ``x[edge_index[1]]`` as ``_i``, reduce messages over ``edge_index[1]``), the
  whole drug side (``GINEConv`` + ``MLP``) and ``to_dense_batch`` live in
  torch_geometric (>=2.5.2 per requirements.txt:2; not vendored, not installed,
  no reference test or golden vector covers them): restated from PyG's
  published definitions -- PARITY UNPINNED at that boundary; pinned only
  structurally by the shapes / key names / parameter counts in
  ``pretrained_model_downstream/`` (see ``tests/test_dropin_boundary.py``).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- a1
def norm_no_nan(x, axis=-1, keepdims=False, eps=1e-8, sqrt=True):
    """gvp_layers.py:79-86 -- L2 norm with the *squared* norm clamped at eps."""
    out = torch.clamp(torch.sum(torch.square(x), axis, keepdims), min=eps)
    return torch.sqrt(out) if sqrt else out


def _act(name):
    if name is None:
        return None
    return {"relu": F.relu, "sigmoid": torch.sigmoid,
            "leaky_relu": lambda t: F.leaky_relu(t, 0.01)}[name]


# ----------------------------------------------------- test-only emulation of reduced-precision matrix-core operands
_GEMM_DTYPE = None


class emulate_gemm_dtype:
    """`with emulate_gemm_dtype(torch.bfloat16):` -- every nn.Linear of a GVP is evaluated the way the bf16-storage
    kernels evaluate it on v_mfma_f32_16x16x16_bf16: operands (input, weight; in the backward also the incoming
    gradient) rounded to bf16, products accumulated in fp32, bias and bias gradient in fp32.  Linears with at most 4
    input channels (one k-step: they stay on the fp32 instruction) are exact in the forward and in the data gradient;
    their weight gradient still uses rounded operands (the item-reduction outer products all run on the bf16
    instruction).  Checker of tests/test_bf16_storage.py only."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _GEMM_DTYPE
        self.prev, _GEMM_DTYPE = _GEMM_DTYPE, self.dtype

    def __exit__(self, *exc):
        global _GEMM_DTYPE
        _GEMM_DTYPE = self.prev
        return False


class _EmuLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, dtype, exact_cols):
        rd = lambda t: t.to(dtype).to(t.dtype)
        narrow = W.shape[1] <= 4
        xr, Wr = rd(x), rd(W)
        if exact_cols:                  # one-hot type columns: the kernels add the fp32 weight column (no product to round)
            Wr = torch.cat([W[:, :exact_cols], Wr[:, exact_cols:]], 1)
        ctx.save_for_backward(W if narrow else Wr, xr)
        ctx.dtype, ctx.narrow, ctx.has_bias = dtype, narrow, b is not None
        return F.linear(x if narrow else xr, W if narrow else Wr, b)

    @staticmethod
    def backward(ctx, gy):
        Wd, xr = ctx.saved_tensors
        gr = gy.to(ctx.dtype).to(gy.dtype)
        gx = (gy if ctx.narrow else gr) @ Wd
        gW = gr.reshape(-1, gr.shape[-1]).t() @ xr.reshape(-1, xr.shape[-1])
        gb = gy.reshape(-1, gy.shape[-1]).sum(0) if ctx.has_bias else None
        return gx, gW, gb, None, None


def _linear(x, W, b=None, exact_cols=0):
    if _GEMM_DTYPE is None:
        return F.linear(x, W, b)
    return _EmuLinear.apply(x, W, b, _GEMM_DTYPE, exact_cols)


# --------------------------------------------------------------------------- a3
def gvp(P, pfx, x, vi, vo, scalar_act="relu", vector_act="sigmoid", vector_gate=False, type_cols=0):
    """gvp_layers.py:142-175.  ``x`` is (s, V) when vi > 0 else s.  (`type_cols`: leading one-hot columns of s; only
    the reduced-precision emulation looks at it.)"""
    sa, va = _act(scalar_act), _act(vector_act)
    if vi:
        s, v = x
        v = torch.transpose(v, -1, -2)                       # :151
        vh = _linear(v, P[pfx + "wh.weight"])               # :152
        vn = norm_no_nan(vh, axis=-2)                        # :153
        s = _linear(torch.cat([s, vn], -1), P[pfx + "ws.weight"], P[pfx + "ws.bias"], type_cols)  # :154
        if vo:
            v = _linear(vh, P[pfx + "wv.weight"])           # :156
            v = torch.transpose(v, -1, -2)                   # :157
            if vector_gate:                                  # :158-163 (gate sees pre-activation s)
                gate_in = va(s) if va is not None else s
                gate = _linear(gate_in, P[pfx + "wsv.weight"], P[pfx + "wsv.bias"])
                v = v * torch.sigmoid(gate).unsqueeze(-1)
            elif va is not None:                             # :164-166
                v = v * va(norm_no_nan(v, axis=-1, keepdims=True))
    else:
        s = _linear(x, P[pfx + "ws.weight"], P[pfx + "ws.bias"])   # :168
        if vo:
            v = torch.zeros(s.shape[0], vo, 3, dtype=s.dtype)        # :170
    if sa is not None:
        s = sa(s)                                            # :172-173
    return (s, v) if vo else s


# --------------------------------------------------------------------------- a4
def gvp_layernorm(P, pfx, x, nv):
    """gvp_layers.py:231-242."""
    w, b = P[pfx + "scalar_norm.weight"], P[pfx + "scalar_norm.bias"]
    if not nv:
        return F.layer_norm(x, w.shape, w, b, 1e-5)
    s, v = x
    vn = norm_no_nan(v, axis=-1, keepdims=True, sqrt=False)
    vn = torch.sqrt(torch.mean(vn, dim=-2, keepdim=True))
    return F.layer_norm(s, w.shape, w, b, 1e-5), v / vn


# --------------------------------------------------------------------------- a6
def gvp_conv(P, pfx, x, edge_index, edge_attr, n_layers=3, aggr="mean",
             activations=("relu", "sigmoid"), vector_gate=False):
    """gvp_layers.py:291-308 + PyG MessagePassing.propagate (restated).

    message_func layout is gvp_layers.py:275-289: first GVP maps the
    concatenated (2*si+se, 2*vi+ve) tuple to out_dims, middle GVPs keep
    out_dims, the last one has activations (None, None).
    """
    s, v = x
    n = s.shape[0]
    nv = v.shape[1]
    src, dst = edge_index[0], edge_index[1]
    s_j, v_j = s.index_select(0, src), v.index_select(0, src)        # x_j = x[edge_index[0]]
    s_i, v_i = s.index_select(0, dst), v.index_select(0, dst)        # x_i = x[edge_index[1]]
    m = (torch.cat([s_j, edge_attr[0], s_i], -1),
         torch.cat([v_j, edge_attr[1], v_i], -2))                    # :306 tuple_cat
    vi_cat = m[1].shape[1]
    for l in range(n_layers):
        last = (l == n_layers - 1)
        sa, va = (None, None) if last else activations
        m = gvp(P, f"{pfx}message_func.{l}.", m, vi_cat if l == 0 else nv, nv, sa, va, vector_gate)
    ms, mv = m
    merged = torch.cat([ms, mv.reshape(mv.shape[0], 3 * nv)], -1)    # :308 _merge
    out = torch.zeros(n, merged.shape[1], dtype=merged.dtype).index_add_(0, dst, merged)
    if aggr == "mean":
        deg = torch.zeros(n, dtype=merged.dtype).index_add_(0, dst, torch.ones_like(dst, dtype=merged.dtype))
        out = out / deg.clamp(min=1).unsqueeze(-1)
    elif aggr not in ("sum", "add"):
        raise ValueError(aggr)
    return out[:, :-3 * nv], out[:, -3 * nv:].reshape(n, nv, 3)      # :301 _split


def _ste_round(t, dtype):
    """Round to `dtype` and back (what storing an activation in that type does), with a straight-through gradient."""
    if dtype is None:
        return t
    if isinstance(t, tuple):
        return tuple(_ste_round(u, dtype) for u in t)
    return t + (t.to(dtype).to(t.dtype) - t).detach()


# --------------------------------------------------------------------------- a7
def _apply_mask(t, mask):
    """gvp_layers.Dropout with a GIVEN mask row [s factors | v-channel factors]
    (gvp_layers.py:187-219: a vector channel's xyz share one factor)."""
    if mask is None:
        return t
    ns = t[0].shape[1]
    return t[0] * mask[:, :ns], t[1] * mask[:, ns:].unsqueeze(-1)


def gvp_conv_layer(P, pfx, x, edge_index, edge_attr, n_message=3, n_feedforward=2,
                   aggr="mean", activations=("relu", "sigmoid"), vector_gate=False, masks=(None, None),
                   store_dtype=None):
    """gvp_layers.py:366-415 without the autoregressive / node_mask branches.
    Eval mode by default (dropout = identity, :192-193); `masks` supplies the two
    dropout masks explicitly to check the training kernels."""
    nv = x[1].shape[1]
    dh = gvp_conv(P, pfx + "conv.", x, edge_index, edge_attr, n_message, aggr, activations, vector_gate)
    dh = _ste_round(dh, store_dtype)          # the aggregated messages are a stored stage hand-off (see protein_lba_forward)
    dh = _apply_mask(dh, masks[0])
    x = gvp_layernorm(P, pfx + "norm.0.", (x[0] + dh[0], x[1] + dh[1]), nv)     # :407
    h = x
    hv = nv
    for l in range(n_feedforward):                                              # :355-364
        last = (l == n_feedforward - 1)
        sa, va = (None, None) if last else activations
        vo = nv if last else 2 * nv
        if n_feedforward == 1:
            vo = nv
        h = gvp(P, f"{pfx}ff_func.{l}.", h, hv, vo, sa, va, vector_gate)
        hv = vo
    h = _apply_mask(h, masks[1])
    return gvp_layernorm(P, pfx + "norm.1.", (x[0] + h[0], x[1] + h[1]), nv)    # :410


# --------------------------------------------------------------------------- a8/a9
def protein_lba_forward(P, x, edge_index, ntypes, etypes, eattr, num_ntypes=20, num_etypes=1,
                        num_convs=2, aggr="sum", pfx="", return_stages=False, masks=None, store_dtype=None):
    """protein_gnn.py:361-388 (VectorProteinGNN_LBAModel.forward), eval mode.

    `store_dtype` (e.g. torch.bfloat16) emulates an implementation that keeps the activations it hands from stage to
    stage in that type: the node embedding, the edge embedding, every layer's aggregated messages and output and the
    result are rounded to it (straight-through gradient) -- the checker of the bf16-storage kernels.

    One-hot node / edge types are concatenated IN FRONT of the scalar features
    (protein_gnn.py:139-152).  GVP activations are (ReLU, None) with
    vector_gate=True (protein_gnn.py:346-358); the two input GVPs and the last
    GVP of each message / feed-forward stack have no activations.
    """
    x_s, x_v = x
    e_s, e_v = eattr
    dt = x_s.dtype
    x_s = torch.cat([F.one_hot(ntypes, num_ntypes).to(dt), x_s], -1)
    e_s = torch.cat([F.one_hot(etypes, num_etypes).to(dt), e_s], -1)
    hv = P[pfx + "gvp_node.0.wv.weight"].shape[0]
    ev = P[pfx + "gvp_edge.0.wv.weight"].shape[0]
    stages = {}
    h = gvp(P, pfx + "gvp_node.0.", (x_s, x_v), x_v.shape[1], hv, None, None, True, num_ntypes)
    h = gvp_layernorm(P, pfx + "gvp_node.1.", h, hv)                            # :375
    e = gvp(P, pfx + "gvp_edge.0.", (e_s, e_v), e_v.shape[1], ev, None, None, True, num_etypes)
    e = _ste_round(gvp_layernorm(P, pfx + "gvp_edge.1.", e, ev), store_dtype)   # :376
    h = _ste_round(h, store_dtype)
    stages["node_embed"] = h
    stages["edge_embed"] = e
    for l in range(num_convs):                                                  # :379-380
        h = gvp_conv_layer(P, f"{pfx}conv_list.{l}.", h, edge_index, e, 3, 2,
                           aggr, ("relu", None), True, masks[l] if masks is not None else (None, None), store_dtype)
        h = _ste_round(h, store_dtype)
        stages[f"conv{l}"] = h
    h = gvp_layernorm(P, pfx + "gvp_norm_before_scalar.", h, hv)                # :385
    out = _ste_round(gvp(P, pfx + "gvp_to_scalar.", h, hv, 0, "relu", None, True), store_dtype)   # :386
    return (out, stages) if return_stages else out


# --------------------------------------------------------------------------- a12/a13
def gine_conv(P, pfx, x, edge_index, edge_attr, act="leaky_relu"):
    """PyG GINEConv(nn=MLP([in,out,out]), train_eps, edge_dim) -- restated from
    the published definition, PARITY UNPINNED (see module docstring):

        x_i' = MLP((1 + eps) * x_i + sum_{j->i} ReLU(x_j + W_e e_ji + b_e))
        MLP  = Linear -> act -> Linear   (norm=None, plain last layer)

    Called from molecule_gnn.py:271-280; parameter names conv_list.{l}.{eps,
    lin.weight, lin.bias, nn.lins.{0,1}.{weight,bias}}.
    """
    src, dst = edge_index[0], edge_index[1]
    e = F.linear(edge_attr, P[pfx + "lin.weight"], P[pfx + "lin.bias"])
    m = F.relu(x.index_select(0, src) + e)
    agg = torch.zeros_like(x).index_add_(0, dst, m)
    h = (1 + P[pfx + "eps"]) * x + agg
    h = F.linear(h, P[pfx + "nn.lins.0.weight"], P[pfx + "nn.lins.0.bias"])
    h = _act(act)(h)
    return F.linear(h, P[pfx + "nn.lins.1.weight"], P[pfx + "nn.lins.1.bias"])


def molecule_gine_forward(P, x, edge_index, ntypes, etypes, eattr, num_ntypes=11, num_etypes=5,
                          num_convs=2, act="leaky_relu", pfx="", return_stages=False, masks=None):
    """molecule_gnn.py:254-268 (HomoMoleculeGNN_GINE.forward), eval mode; `masks` (one [N, width] tensor of
    dropout factors per layer but the last, or None) supplies the training-mode dropout of :262 explicitly.
    Type encoders (:112-122): nn.Embedding tables when P holds `ntype_embedding.weight` / `etype_embedding.weight`,
    one-hot otherwise."""
    dt = x.dtype
    nt = P[pfx + "ntype_embedding.weight"][ntypes] if pfx + "ntype_embedding.weight" in P else F.one_hot(ntypes, num_ntypes).to(dt)
    et = P[pfx + "etype_embedding.weight"][etypes] if pfx + "etype_embedding.weight" in P else F.one_hot(etypes, num_etypes).to(dt)
    x = torch.cat([nt, x], -1)                                                  # :127-140
    eattr = torch.cat([et, eattr], -1)
    stages = {}
    for l in range(num_convs):
        x = _act(act)(gine_conv(P, f"{pfx}conv_list.{l}.", x, edge_index, eattr, act))
        if masks is not None and l < num_convs - 1 and masks[l] is not None:
            x = x * masks[l]                                                    # :262 dropout between layers
        stages[f"conv{l}"] = x
    return (x, stages) if return_stages else x


# --------------------------------------------------------------------------- head
def to_dense_batch(x, batch, num_graphs=None):
    """PyG utils.to_dense_batch restated (PARITY UNPINNED): pad each graph's
    rows to the longest graph -> ([B, Lmax, D], bool mask [B, Lmax])."""
    if num_graphs is None:
        num_graphs = int(batch.max()) + 1 if batch.numel() else 0
    counts = torch.bincount(batch, minlength=num_graphs)
    lmax = int(counts.max()) if counts.numel() else 0
    ptr = torch.cumsum(counts, 0) - counts
    pos = torch.arange(x.shape[0]) - ptr[batch]
    out = x.new_zeros(num_graphs, lmax, x.shape[1])
    mask = torch.zeros(num_graphs, lmax, dtype=torch.bool)
    out[batch, pos] = x
    mask[batch, pos] = True
    return out, mask


def _mha(P, pfx, q, kv, key_padding_mask, heads):
    """nn.MultiheadAttention(batch_first=True, same q/k/v dim) forward, eval."""
    return F.multi_head_attention_forward(
        q.transpose(0, 1), kv.transpose(0, 1), kv.transpose(0, 1), q.shape[-1], heads,
        P[pfx + "in_proj_weight"], P[pfx + "in_proj_bias"], None, None, False, 0.0,
        P[pfx + "out_proj.weight"], P[pfx + "out_proj.bias"], training=False,
        key_padding_mask=key_padding_mask, need_weights=False)[0].transpose(0, 1)


def joint_head_forward(P, residue_embed, atom_embed, pbatch, mbatch, heads=8, act="leaky_relu"):
    """joint_gnn.py:188-286 for the shipped configuration (depth-1 linears,
    one cross-attention layer with residual stream, mean pooling), eval mode."""
    a = _act(act)
    ln = lambda t, k: F.layer_norm(t, (t.shape[-1],), P[k + ".weight"], P[k + ".bias"], 1e-5)
    lin = lambda t, k: F.linear(t, P[k + ".weight"], P[k + ".bias"])
    r = a(lin(residue_embed, "residue_lins.0"))                                 # :188-192
    m = a(lin(atom_embed, "atom_lins.0"))                                       # :194-198
    r, rmask = to_dense_batch(r, pbatch)                                        # :206
    m, mmask = to_dense_batch(m, mbatch)                                        # :207
    c = "cross_attn_module.cross_attn_layers.0."
    r_n, m_n = ln(r, c + "preattn_norm1"), ln(m, c + "preattn_norm2")           # :376-377
    r_att = _mha(P, c + "embed1_to_2.", r_n, m_n, ~mmask, heads)                # :379
    m_att = _mha(P, c + "embed2_to_1.", m_n, r_n, ~rmask, heads)                # :380
    r = r + r_att                                                               # :384
    r = r + lin(F.relu(lin(ln(r, c + "ff_norm1"), c + "ff1.0")), c + "ff1.3")   # :386-389
    m = m + m_att
    m = m + lin(F.relu(lin(ln(m, c + "ff_norm2"), c + "ff2.0")), c + "ff2.3")
    pe = (r * rmask.unsqueeze(-1)).sum(1) / rmask.sum(1, keepdim=True)          # :225
    me = (m * mmask.unsqueeze(-1)).sum(1) / mmask.sum(1, keepdim=True)          # :226
    pe, me = a(pe), a(me)                                                       # :249-253
    pe = a(lin(pe, "protein_lins.0"))                                           # :257-261
    me = a(lin(me, "molecule_lins.0"))
    pair = torch.cat([pe, me], -1)                                              # :272 (all-gather point)
    z = a(lin(pair, "pm_embed_lin"))                                            # :273-274
    z = a(lin(z, "out_fc_layers.0"))                                            # :279-283
    return lin(z, "output_layer"), pair                                         # :286


def joint_forward(P, pdata, mdata):
    """joint_gnn.py:172-288 end to end with the shipped kwargs."""
    pp = {k[len("protein_gnn.gnn_model."):]: v for k, v in P.items() if k.startswith("protein_gnn.gnn_model.")}
    mp = {k[len("molecule_gnn.gnn_model."):]: v for k, v in P.items() if k.startswith("molecule_gnn.gnn_model.")}
    res = protein_lba_forward(pp, pdata["x"], pdata["edge_index"], pdata["ntypes"], pdata["etypes"], pdata["eattr"])
    atm = molecule_gine_forward(mp, mdata["x"], mdata["edge_index"], mdata["ntypes"], mdata["etypes"], mdata["eattr"])
    return joint_head_forward(P, res, atm, pdata["batch"], mdata["batch"])[0]
