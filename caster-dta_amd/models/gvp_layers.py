"""`models.gvp_layers` with the reference's public surface (gvp_layers.py:39-415):
tuple helpers, `GVP`, `LayerNorm`, `Dropout`, `GVPConv`, `GVPConvLayer`, with the
same constructor signatures and the same parameter names / shapes, so existing
checkpoints load with strict=True.

Where the arithmetic runs
-------------------------
The hot path never executes the `forward` methods in this file: the protein
encoder (`models.protein_gnn.VectorProteinGNN_LBAModel`) walks these modules only
to own their Parameters (as views into one arena) and launches the fused gfx950
kernels of libcaster_gvp.so.

A `GVPConv` / `GVPConvLayer` instantiated on its own (the PocketMiner- and
CPD-style stacks, analysis scripts) runs on the same tile kernels when it is one
of the three layer kinds they are compiled for -- node dims (16, 4), edge dims
(<= 32, 1), 3 message / 2 feed-forward GVPs, (relu, None)+gate, (relu, sigmoid)
or (None, None) -- and its tensors are fp32 on the GPU (`gvp_hip.conv_layer_ops`).
Every other case runs the small tensor-op compositions below (no torch_geometric
needed), which are also what the golden fixtures pin against the reference.
"""
import functools

import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------- tuple helpers
def tuple_sum(*args):
    """Elementwise sum of (s, V) tuples."""
    return tuple(sum(parts) for parts in zip(*args))


def tuple_cat(*args, dim=-1):
    """Concatenate (s, V) tuples; `dim` counts on the scalar tensor, so the
    vector tensor (one extra trailing xyz axis) uses dim-1 when dim is negative."""
    dim %= args[0][0].dim()
    return (torch.cat([a[0] for a in args], dim=dim), torch.cat([a[1] for a in args], dim=dim))


def tuple_index(x, idx):
    return x[0][idx], x[1][idx]


def randn(n, dims, device="cpu"):
    return torch.randn(n, dims[0], device=device), torch.randn(n, dims[1], 3, device=device)


def _norm_no_nan(x, axis=-1, keepdims=False, eps=1e-8, sqrt=True):
    """L2 norm whose SQUARE is clamped at eps (so zero vectors give 1e-4, not NaN grads)."""
    sq = x.square().sum(axis, keepdims).clamp(min=eps)
    return sq.sqrt() if sqrt else sq


def _split(x, nv):
    """[..., s + 3*nv] -> ([..., s], [..., nv, 3])."""
    return x[..., :x.shape[-1] - 3 * nv], x[..., x.shape[-1] - 3 * nv:].reshape(*x.shape[:-1], nv, 3)


def _merge(s, v):
    """([..., s], [..., nv, 3]) -> [..., s + 3*nv] (the kernels' node row layout)."""
    return torch.cat([s, v.reshape(*v.shape[:-2], 3 * v.shape[-2])], -1)


# --------------------------------------------------------------------- modules
class GVP(nn.Module):
    """Geometric vector perceptron: (s, V) -> (s', V').

    Parameters (names fixed by the checkpoint format): `wh` [h, vi] mixes vector
    channels, `ws` [so, si+h] maps scalars ++ channel norms, `wv` [vo, h],
    `wsv` [vo, so] (vector gate), plus a zero-size `dummy_param`.
    """

    def __init__(self, in_dims, out_dims, h_dim=None, activations=(F.relu, torch.sigmoid), vector_gate=False):
        super().__init__()
        self.si, self.vi = in_dims
        self.so, self.vo = out_dims
        self.vector_gate = vector_gate
        if self.vi:
            self.h_dim = h_dim or max(self.vi, self.vo)
            self.wh = nn.Linear(self.vi, self.h_dim, bias=False)
            self.ws = nn.Linear(self.h_dim + self.si, self.so)
            if self.vo:
                self.wv = nn.Linear(self.h_dim, self.vo, bias=False)
                if vector_gate:
                    self.wsv = nn.Linear(self.so, self.vo)
        else:
            self.ws = nn.Linear(self.si, self.so)
        self.scalar_act, self.vector_act = activations
        self.dummy_param = nn.Parameter(torch.empty(0))

    def forward(self, x):
        if not self.vi:
            s = self.ws(x)
            v = s.new_zeros(s.shape[0], self.vo, 3) if self.vo else None
        else:
            s, v = x
            vh = torch.einsum("hi,...ic->...hc", self.wh.weight, v)
            s = self.ws(torch.cat([s, _norm_no_nan(vh, axis=-1)], -1))
            if self.vo:
                v = torch.einsum("oh,...hc->...oc", self.wv.weight, vh)
                if self.vector_gate:            # gate reads the PRE-activation scalars
                    g = self.vector_act(s) if self.vector_act else s
                    v = v * torch.sigmoid(self.wsv(g)).unsqueeze(-1)
                elif self.vector_act:
                    v = v * self.vector_act(_norm_no_nan(v, axis=-1, keepdims=True))
        if self.scalar_act:
            s = self.scalar_act(s)
        return (s, v) if self.vo else s


class _VDropout(nn.Module):
    """Drops whole vector channels (the three xyz components share one mask)."""

    def __init__(self, drop_rate):
        super().__init__()
        self.drop_rate = drop_rate
        self.dummy_param = nn.Parameter(torch.empty(0))

    def forward(self, x):
        if not self.training:
            return x
        keep = 1.0 - self.drop_rate
        mask = torch.bernoulli(torch.full(x.shape[:-1], keep, device=x.device, dtype=x.dtype))
        return x * mask.unsqueeze(-1) / keep


class Dropout(nn.Module):
    def __init__(self, drop_rate):
        super().__init__()
        self.sdropout = nn.Dropout(drop_rate)
        self.vdropout = _VDropout(drop_rate)

    def forward(self, x):
        if isinstance(x, torch.Tensor):
            return self.sdropout(x)
        return self.sdropout(x[0]), self.vdropout(x[1])


class LayerNorm(nn.Module):
    """nn.LayerNorm on the scalars; vectors divided by the RMS of their
    (clamped) channel norms -- no learnable parameters on the vector side."""

    def __init__(self, dims):
        super().__init__()
        self.s, self.v = dims
        self.scalar_norm = nn.LayerNorm(self.s)

    def forward(self, x):
        if not self.v:
            return self.scalar_norm(x)
        s, v = x
        rms = _norm_no_nan(v, axis=-1, keepdims=True, sqrt=False).mean(dim=-2, keepdim=True).sqrt()
        return self.scalar_norm(s), v / rms


def _gather_scatter(message_fn, x_s, x_v, edge_index, edge_attr, aggr):
    """PyG `MessagePassing.propagate` semantics for flow source->target:
    `_j` = rows at edge_index[0], `_i` = rows at edge_index[1]; messages are
    reduced over edge_index[1] (sum/add or mean with in-degree clamped at 1)."""
    src, dst = edge_index[0], edge_index[1]
    msg = message_fn((x_s.index_select(0, dst), x_v.index_select(0, dst)),
                     (x_s.index_select(0, src), x_v.index_select(0, src)), edge_attr)
    out = msg.new_zeros(x_s.shape[0], msg.shape[1]).index_add_(0, dst, msg)
    if aggr == "mean":
        deg = torch.bincount(dst, minlength=x_s.shape[0]).clamp(min=1).to(msg.dtype)
        out = out / deg.unsqueeze(-1)
    elif aggr not in ("add", "sum"):
        raise ValueError(f"unsupported aggregation {aggr!r}")
    return out


class GVPConv(nn.Module):
    """Message passing with a stack of GVPs as message function; no residual,
    no feed-forward (that is `GVPConvLayer`)."""

    def __init__(self, in_dims, out_dims, edge_dims, n_layers=3, module_list=None, aggr="mean",
                 activations=(F.relu, torch.sigmoid), vector_gate=False):
        super().__init__()
        self.aggr = aggr
        self.si, self.vi = in_dims
        self.so, self.vo = out_dims
        self.se, self.ve = edge_dims
        make = functools.partial(GVP, activations=activations, vector_gate=vector_gate)
        cat_dims = (2 * self.si + self.se, 2 * self.vi + self.ve)
        stack = list(module_list or [])
        if not stack:
            if n_layers == 1:
                stack = [make(cat_dims, (self.so, self.vo), activations=(None, None))]
            else:
                stack = [make(cat_dims, out_dims)]
                stack += [make(out_dims, out_dims) for _ in range(n_layers - 2)]
                stack += [make(out_dims, out_dims, activations=(None, None))]
        self.message_func = nn.Sequential(*stack)

    def message(self, x_i, x_j, edge_attr):
        return _merge(*self.message_func(tuple_cat(x_j, edge_attr, x_i)))

    def _kernel_kind(self, x, edge_attr):
        """The layer kind the MI355X kernels compute this conv as (gvp_hip.conv_layer_ops), or None: other widths,
        other activation combinations, non-CUDA / non-fp32 tensors run the tensor-op composition below."""
        if not (x[0].is_cuda and x[0].dtype == torch.float32):
            return None
        from gvp_hip import conv_layer_ops as K
        if not K.usable(x[0], x[1], edge_attr[0], edge_attr[1]):
            return None
        return K.conv_kind(self)

    def forward(self, x, edge_index, edge_attr):
        kind = self._kernel_kind(x, edge_attr)
        if kind is not None:
            from gvp_hip import conv_layer_ops as K
            return K.tuple_from_rows(K.conv_message(self, kind, x, edge_index, edge_attr))
        agg = _gather_scatter(self.message, x[0], x[1], edge_index, edge_attr, self.aggr)
        return _split(agg, self.vo)


class GVPConvLayer(nn.Module):
    """GVPConv + residual/LayerNorm + pointwise GVP feed-forward + residual/LayerNorm."""

    def __init__(self, node_dims, edge_dims, n_message=3, n_feedforward=2, drop_rate=.1, autoregressive=False,
                 activations=(F.relu, torch.sigmoid), vector_gate=False, aggr=None):
        super().__init__()
        if autoregressive:
            if aggr is not None and aggr != "add":
                raise ValueError("Cannot use autoregressive and aggr together in GVPConvLayer "
                                 "unless aggr is set to 'add'")
            aggr = "add"
        elif aggr is None:
            aggr = "mean"
        self.conv = GVPConv(node_dims, node_dims, edge_dims, n_message, aggr=aggr,
                            activations=activations, vector_gate=vector_gate)
        make = functools.partial(GVP, activations=activations, vector_gate=vector_gate)
        self.norm = nn.ModuleList([LayerNorm(node_dims) for _ in range(2)])
        self.dropout = nn.ModuleList([Dropout(drop_rate) for _ in range(2)])
        if n_feedforward == 1:
            ff = [make(node_dims, node_dims, activations=(None, None))]
        else:
            hid = (4 * node_dims[0], 2 * node_dims[1])
            ff = [make(node_dims, hid)]
            ff += [make(hid, hid) for _ in range(n_feedforward - 2)]
            ff += [make(hid, node_dims, activations=(None, None))]
        self.ff_func = nn.Sequential(*ff)

    def forward(self, x, edge_index, edge_attr, autoregressive_x=None, node_mask=None):
        if autoregressive_x is None:
            dh = self.conv(x, edge_index, edge_attr)
        else:                                   # messages with src >= dst read `autoregressive_x`
            fwd = edge_index[0] < edge_index[1]
            parts = []
            for sel, feats in ((fwd, x), (~fwd, autoregressive_x)):
                parts.append(self.conv(feats, edge_index[:, sel], tuple_index(edge_attr, sel)))
            dh = tuple_sum(*parts)
            n = dh[0].shape[0]
            cnt = torch.bincount(edge_index[1], minlength=n).clamp(min=1).to(dh[0].dtype).unsqueeze(-1)
            dh = (dh[0] / cnt, dh[1] / cnt.unsqueeze(-1))
        if x[0].is_cuda and x[0].dtype == torch.float32:
            from gvp_hip import conv_layer_ops as K
            kind = K.node_kind(self) if K.usable(x[0], x[1], dh[0], dh[1]) else None
            if kind is not None:            # residual + LayerNorm + feed-forward + residual + LayerNorm: one launch
                if node_mask is None:
                    return K.node_update(self, kind, x, K.rows_from_tuple(dh))
                # only the masked nodes are updated (gvp_layers.py:403-414): the kernel runs on their rows, the
                # result is written back into the caller's tensors like the reference does
                sub = K.node_update(self, kind, tuple_index(x, node_mask), K.rows_from_tuple(tuple_index(dh, node_mask)))
                x[0][node_mask], x[1][node_mask] = sub[0], sub[1]
                return x
        full = x
        if node_mask is not None:
            x, dh = tuple_index(x, node_mask), tuple_index(dh, node_mask)
        x = self.norm[0](tuple_sum(x, self.dropout[0](dh)))
        x = self.norm[1](tuple_sum(x, self.dropout[1](self.ff_func(x))))
        if node_mask is not None:
            full[0][node_mask], full[1][node_mask] = x[0], x[1]
            x = full
        return x
