"""Stand-alone `GVPConv` / `GVPConvLayer` on the MI355X tile kernels (survey rows a7 / a11 / f-4).

The protein encoder of CASTER-DTA drives the conv / node-update kernels through the whole-pass entry points.  The
reference builds two more stacks from the same layer class (protein_gnn.py:392-516 PocketMiner-style, :518-608 CPD-style
with an autoregressive decoder) whose layers differ in `activations` / `vector_gate` (gvp_layers.py:340-366):

    kind     activations          vector_gate   vector outputs              used by
    GATED    (relu, None)         True          v * sigmoid(wsv(s))         CASTER-DTA / LBA   protein_gnn.py:349-353
    GVPDEF   (relu, sigmoid)      False         v * sigmoid(|v|)            CPD enc / dec      protein_gnn.py:565-573
    LINEAR   (None, None)         False         v                           PocketMiner        protein_gnn.py:468-473

The kernels are compiled for node dims (16, 4) and an edge embedding of (32, 1); a layer with FEWER edge scalars runs on
them with zero-padded edge rows and zero weight columns (exact: the padded products are 0).  A layer with MORE edge scalars
than 32 -- the default CPD decoder: 32 + 20 (protein_gnn.py:566-570) -- runs on them FOLDED: the edge scalars enter the
message only through message_func.0's `ws`, a linear map onto its 16 pre-activations, so

    ws . [s_j | e (se) | s_i | vn]  =  ws' . [s_j | t (16) 0 (16) | s_i | vn],     t = e . ws[:, 16:16+se]^T,  ws' = ws with the edge
                                                                                  columns replaced by [I_16 | 0]

i.e. the kernels get the 16-wide projection `t` as their "edge scalars" and an identity block as those weight columns; the
E x se x 16 projection and its gradients (d ws_e = d t^T e, d e = d t ws_e) are torch ops around them, composed by autograd.
Everything else about the layer is the kernels' native case: 3 message GVPs, 2 feed-forward GVPs, aggregation mean / add.

How a layer runs here (all through fine-grained C-ABI entry points, include/caster_gvp.h):

    arena   the layer's parameters packed into a ONE-conv-layer arena of the library's layout (torch.cat, differentiable:
            autograd carries the kernels' gradient arena back to the individual nn.Parameters; wsv slots of the
            un-gated kinds are zeros)
    image   cgvp_lba_prepare                                  1 launch, shared by the conv and the node update
    conv    cgvp_conv_fwd (stored edge rows)  /  cgvp_conv_bwd + cgvp_bwd_reduce
    node    cgvp_node_update_fwd_train        /  cgvp_node_update_bwd + cgvp_bwd_reduce

The autoregressive form (gvp_layers.py:382-398) is two conv calls over the two edge subsets (src < dst reads `x`, the
rest `autoregressive_x`) with aggregation 'add' and an in-degree divide -- exactly how the reference composes it.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib, ops
from .ops import EROW, ROW, _ptr, _stream

NS, NV, ES, EV = 16, 4, 32, 1
MROW = 20                       # dropout mask row: 16 scalar factors + 4 vector-channel factors
ENABLED = True                  # tests flip this to compare against the tensor-op composition on the same device


def _is_relu(fn):
    return fn is F.relu or fn is torch.relu or isinstance(fn, torch.nn.ReLU)


def gvp_kind(g):
    """cgvp_dims.layer_kind of a first-position GVP (message_func.0 / ff_func.0), or None when the kernels have no
    such variant."""
    if g.vector_gate:
        return ops.LAYER_GATED if (_is_relu(g.scalar_act) and g.vector_act is None) else None
    if _is_relu(g.scalar_act) and g.vector_act is torch.sigmoid:
        return ops.LAYER_GVPDEF
    if g.scalar_act is None and g.vector_act is None:
        return ops.LAYER_LINEAR
    return None


def _last_ok(g, kind):
    """message_func.2 / ff_func.1: activations (None, None), same gate setting."""
    return g.scalar_act is None and g.vector_act is None and bool(g.vector_gate) == (kind == ops.LAYER_GATED)


def conv_kind(conv):
    """Layer kind if `conv` (a GVPConv) is a case the kernels compute, else None."""
    mf = list(conv.message_func)
    if (conv.si, conv.vi, conv.so, conv.vo) != (NS, NV, NS, NV) or conv.ve != EV or conv.se < 0:
        return None          # (any number of edge scalars: <= 32 padded, more than 32 folded through message_func.0's ws)
    if len(mf) != 3 or conv.aggr not in ("mean", "add", "sum"):
        return None
    dims = [((2 * NS + conv.se, 2 * NV + EV), (NS, NV)), ((NS, NV), (NS, NV)), ((NS, NV), (NS, NV))]
    for g, (i, o) in zip(mf, dims):
        if (g.si, g.vi, g.so, g.vo) != (*i, *o) or g.h_dim != max(i[1], o[1]):
            return None
    kind = gvp_kind(mf[0])
    if kind is None or gvp_kind(mf[1]) != kind or not _last_ok(mf[2], kind):
        return None
    return kind


def node_kind(layer):
    """Layer kind if the residual / feed-forward half of `layer` (a GVPConvLayer) is a case the kernels compute."""
    ff = list(layer.ff_func)
    if len(ff) != 2 or (layer.norm[0].s, layer.norm[0].v) != (NS, NV):
        return None
    dims = [((NS, NV), (4 * NS, 2 * NV)), ((4 * NS, 2 * NV), (NS, NV))]
    for g, (i, o) in zip(ff, dims):
        if (g.si, g.vi, g.so, g.vo) != (*i, *o) or g.h_dim != max(i[1], o[1]):
            return None
    kind = gvp_kind(ff[0])
    if kind is None or not _last_ok(ff[1], kind):
        return None
    return kind


def usable(*tensors):
    return ENABLED and all(t.is_cuda and t.dtype == torch.float32 for t in tensors)


_CFG = {}


def _cfg(kind):
    """(dims, layout) of a one-layer arena without type columns, for the given layer kind."""
    if kind not in _CFG:
        dims = ops.make_dims(layer_kind=kind)
        _CFG[kind] = (dims, ops.lba_layout(dims, 0, 0, 1))
    return _CFG[kind]


def folds_edges(conv):
    """More edge scalars than the compiled width: the conv runs with the edge scalars projected by message_func.0."""
    return conv.se > ES


def _gvp_block(g, ref, pad=None, fold=None):
    """One GVP's arena block [wh | ws.weight | ws.bias | wv | wsv.weight | wsv.bias] (zeros for an absent gate).
    `pad` = (first column, count): zero columns inserted into ws.weight (edge scalars padded to the compiled width);
    `fold` = (first column, count): those columns (too many edge scalars) replaced by [I | 0] of the compiled width."""
    ws = g.ws.weight
    if fold is not None:
        eye = torch.eye(ws.shape[0], ES, dtype=ws.dtype, device=ws.device)
        ws = torch.cat([ws[:, :fold[0]], eye, ws[:, fold[0] + fold[1]:]], dim=1)
    elif pad is not None and pad[1] > 0:
        ws = torch.cat([ws[:, :pad[0]], ws.new_zeros(ws.shape[0], pad[1]), ws[:, pad[0]:]], dim=1)
    parts = [g.wh.weight.reshape(-1), ws.reshape(-1), g.ws.bias, g.wv.weight.reshape(-1)]
    if g.vector_gate:
        parts += [g.wsv.weight.reshape(-1), g.wsv.bias]
    else:
        parts.append(ref.new_zeros(g.vo * g.so + g.vo))
    return parts


def pack_arena(kind, conv=None, layer=None):
    """The one-layer arena (include/caster_gvp.h, "PARAMETER ARENA") holding `conv`'s message function and / or
    `layer`'s norms + feed-forward; blocks that are not given are zeros.  Differentiable."""
    _, layout = _cfg(kind)
    ref = (conv.message_func[0].ws.weight if conv is not None else layer.ff_func[0].ws.weight)
    parts = [ref.new_zeros(layout.conv0)]
    n_msg = 1369 + 2 * 436
    n_norm = 4 * NS
    if conv is not None:
        if folds_edges(conv):
            parts += _gvp_block(conv.message_func[0], ref, fold=(NS, conv.se))
        else:
            parts += _gvp_block(conv.message_func[0], ref, pad=(NS + conv.se, ES - conv.se))
        parts += _gvp_block(conv.message_func[1], ref) + _gvp_block(conv.message_func[2], ref)
    else:
        parts.append(ref.new_zeros(n_msg))
    if layer is not None:
        for n in layer.norm:
            parts += [n.scalar_norm.weight, n.scalar_norm.bias]
        parts += _gvp_block(layer.ff_func[0], ref) + _gvp_block(layer.ff_func[1], ref)
        tail = layout.total - layout.conv0 - layout.conv_stride
    else:
        tail = layout.total - layout.conv0 - n_msg
    parts.append(ref.new_zeros(tail))
    arena = torch.cat(parts)
    if arena.numel() != layout.total:
        raise RuntimeError(f"packed arena has {arena.numel()} floats, the library's layout {layout.total}")
    return arena


def prepare(kind, arena):
    dims, layout = _cfg(kind)
    return ops.prepare_image(arena.detach(), layout, dims)


def rows_from_tuple(x):
    s, v = x
    return torch.cat([s, v.reshape(v.shape[0], 3 * NV)], dim=1).contiguous()


def tuple_from_rows(r):
    return r[:, :NS], r[:, NS:].reshape(r.shape[0], NV, 3)


def edge_rows(edge_attr, eperm, num_edges, conv=None):
    """Stored edge-embedding rows [E + 1][CGVP_EDGE_ROW] in sorted-edge order (32 scalars zero-padded | xyz | pad; the
    extra last row is zeros).  Differentiable gather.  `conv` with more than 32 edge scalars: the scalars are replaced by
    their projection through message_func.0's edge columns (module docstring, "FOLDED")."""
    e_s, e_v = edge_attr
    if conv is not None and folds_edges(conv):
        e_s = e_s @ conv.message_func[0].ws.weight[:, NS:NS + conv.se].t()
    idx = eperm[:num_edges].long().clamp_min(0)      # (-1: positions behind the last valid edge when the build dropped edges; never read)
    rows = torch.cat([e_s.index_select(0, idx), e_s.new_zeros(num_edges, ES - e_s.shape[1]),
                      e_v.index_select(0, idx).reshape(num_edges, 3), e_s.new_zeros(num_edges, EROW - ES - 3)], dim=1)
    return torch.cat([rows, rows.new_zeros(1, EROW)], dim=0).contiguous()


def _reduce(L, segs, nseg, gparams, st):
    if nseg:
        _lib.check(L.cgvp_bwd_reduce(segs, nseg, _ptr(gparams), 1, st), "cgvp_bwd_reduce")


def _workspace(dims, layout, dev):
    n = int(_lib.lib().cgvp_bwd_workspace_floats(C.byref(dims), C.byref(layout)))
    if n < 0:
        _lib.check(n, "cgvp_bwd_workspace_floats")
    return torch.empty(n, dtype=torch.float32, device=dev)


class _ConvFn(torch.autograd.Function):
    """dh = aggregate of the 3-GVP messages over the CSR (gvp_layers.py:290-308)."""

    @staticmethod
    def forward(ctx, arena, h, e_rows, image, rowptr, esrc, edst, kind, mean):
        L = _lib.lib()
        dims, layout = _cfg(kind)
        N, E = h.shape[0], e_rows.shape[0] - 1
        dh = torch.empty(N, ROW, dtype=torch.float32, device=h.device)
        with torch.cuda.device(h.device):
            _lib.check(L.cgvp_conv_fwd(C.byref(dims), C.byref(layout), _ptr(arena), _ptr(image), 0, _ptr(h), None, None,
                                       None, _ptr(rowptr), None, _ptr(esrc), _ptr(edst), N, E, 1 if mean else 0,
                                       _ptr(e_rows), None, _ptr(dh), _stream()), "cgvp_conv_fwd")
        ctx.save_for_backward(h, e_rows, image, rowptr, esrc, edst)
        ctx.kind, ctx.mean = kind, mean
        return dh

    @staticmethod
    def backward(ctx, g_dh):
        h, e_rows, image, rowptr, esrc, edst = ctx.saved_tensors
        L = _lib.lib()
        dims, layout = _cfg(ctx.kind)
        N, E = h.shape[0], e_rows.shape[0] - 1
        f32 = dict(dtype=torch.float32, device=h.device)
        g_dh = g_dh.contiguous()
        g_src, g_dst, g_e = torch.empty(N, ROW, **f32), torch.empty(N, ROW, **f32), torch.zeros(E + 1, EROW, **f32)
        gparams = torch.zeros(layout.total, **f32)
        segs, cnt = (_lib.Segment * 2)(), C.c_int32(0)
        with torch.cuda.device(h.device):
            st = _stream()
            ws = _workspace(dims, layout, h.device)
            _lib.check(L.cgvp_conv_bwd(C.byref(dims), C.byref(layout), _ptr(image), 0, _ptr(h), _ptr(e_rows),
                                       _ptr(rowptr), _ptr(esrc), _ptr(edst), N, E, 1 if ctx.mean else 0, _ptr(g_dh),
                                       _ptr(g_src), 0, _ptr(g_dst), _ptr(g_e), _ptr(gparams), _ptr(ws), segs,
                                       C.byref(cnt), st), "cgvp_conv_bwd")
            _reduce(L, segs, cnt.value, gparams, st)
        return gparams, g_src + g_dst, g_e, None, None, None, None, None, None


class _NodeFn(torch.autograd.Function):
    """h' = LN(h1 + drop1(ff(h1))), h1 = LN(h + drop0(dh)) (gvp_layers.py:403-408)."""

    @staticmethod
    def forward(ctx, arena, h, dh, image, mask0, mask1, kind):
        L = _lib.lib()
        dims, layout = _cfg(kind)
        N = h.shape[0]
        out = torch.empty(N, ROW, dtype=torch.float32, device=h.device)
        with torch.cuda.device(h.device):
            _lib.check(L.cgvp_node_update_fwd_train(C.byref(dims), C.byref(layout), _ptr(image), 0, _ptr(h), _ptr(dh),
                                                    _ptr(mask0), _ptr(mask1), None, N, 0, _ptr(out), None, _stream()),
                       "cgvp_node_update_fwd_train")
        ctx.save_for_backward(h, dh, image, mask0, mask1)
        ctx.kind = kind
        return out

    @staticmethod
    def backward(ctx, g_out):
        h, dh, image, mask0, mask1 = ctx.saved_tensors
        L = _lib.lib()
        dims, layout = _cfg(ctx.kind)
        N = h.shape[0]
        f32 = dict(dtype=torch.float32, device=h.device)
        g_out = g_out.contiguous()
        g_dh, g_h = torch.empty(N, ROW, **f32), torch.empty(N, ROW, **f32)
        gparams = torch.zeros(layout.total, **f32)
        segs, cnt = (_lib.Segment * 2)(), C.c_int32(0)
        with torch.cuda.device(h.device):
            st = _stream()
            ws = _workspace(dims, layout, h.device)
            _lib.check(L.cgvp_node_update_bwd(C.byref(dims), C.byref(layout), _ptr(image), 0, _ptr(h), _ptr(dh),
                                              _ptr(mask0), _ptr(mask1), None, None, None, _ptr(g_out), None, None, N, 0,
                                              _ptr(g_dh), _ptr(g_h), None, _ptr(gparams), _ptr(ws), segs, C.byref(cnt),
                                              st), "cgvp_node_update_bwd")
            _reduce(L, segs, cnt.value, gparams, st)
        return gparams, g_h, g_dh, None, None, None, None


def _aligned(t):
    t = t.contiguous()
    return t.clone() if t.data_ptr() % 16 else t


def conv_message(conv, kind, x, edge_index, edge_attr, arena=None, image=None):
    """GVPConv.forward on the kernels: -> aggregated messages as node rows [N][28]."""
    h = _aligned(rows_from_tuple(x))
    N, E = h.shape[0], int(edge_index.shape[1])
    if E == 0 or N == 0:
        return h.new_zeros(N, ROW)
    if arena is None:
        arena = pack_arena(kind, conv=conv)
        image = prepare(kind, arena)
    csr = ops.build_csr(edge_index, N)
    e_rows = edge_rows(edge_attr, csr.eperm, E, conv)
    return _ConvFn.apply(arena, h, e_rows, image, csr.rowptr, csr.esrc, csr.edst, kind, conv.aggr == "mean")


def draw_masks(num_nodes, p, device):
    """The factors of gvp_layers.Dropout (gvp_layers.py:187-219) for one use: [N][16 scalar | 4 vector-channel], each
    0 or 1 / (1 - p) (vector channels are dropped whole)."""
    keep = 1.0 - p
    return torch.bernoulli(torch.full((num_nodes, MROW), keep, device=device)) / keep


def node_update(layer, kind, x, dh_rows, arena=None, image=None, masks=None):
    """The residual / LayerNorm / feed-forward half of GVPConvLayer.forward on the kernels; `masks` = (mask0, mask1) to
    use instead of drawing them (None entries = no dropout)."""
    h = _aligned(rows_from_tuple(x))
    if h.shape[0] == 0:
        return x
    if arena is None:
        arena = pack_arena(kind, layer=layer)
        image = prepare(kind, arena)
    if masks is None:
        p = layer.dropout[0].sdropout.p
        masks = (draw_masks(h.shape[0], p, h.device), draw_masks(h.shape[0], p, h.device)) \
            if (layer.training and p > 0) else (None, None)
    out = _NodeFn.apply(arena, h, _aligned(dh_rows), image, masks[0], masks[1], kind)
    return tuple_from_rows(out)
