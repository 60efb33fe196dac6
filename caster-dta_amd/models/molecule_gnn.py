"""`models.molecule_gnn`: the drug-graph encoder behind the reference's module
API (molecule_gnn.py:13-70 wrapper, :73-140 base, :208-280 GINE model).

The reference builds its layers from torch_geometric (`GINEConv(nn=MLP([in, out,
out]), train_eps=True, edge_dim=...)`); torch_geometric is not a dependency here.
`GINEConv` / `MLP` below are parameter containers with PyG's attribute names, so
checkpoints keep their keys (`conv_list.{l}.eps`, `.lin.{weight,bias}`,
`.nn.lins.{0,1}.{weight,bias}`); the arithmetic of a whole layer

    x'_i = act( Lin1( act( Lin0( (1 + eps) x_i + sum_{j->i} ReLU(x_j + W_e e_ji + b_e) ) ) ) )

runs as ONE launch of libcaster_gvp.so per layer (16-atom MFMA tiles, csrc/gine_quad_kernels.hip), the whole encoder
pass as one host call (cgvp_gine_forward_pass).  No eager/CPU fallback.
"""
import warnings
from functools import partial

import torch
import torch.nn as nn

from gvp_hip import ops
from models.model_utils import _select_activation, activation_slope


class SelectableMoleculeModelWrapper(nn.Module):
    """Picks the molecule encoder by `base_conv` (or takes `force_model`, a class);
    forwards calls and unknown attributes to the wrapped model."""

    def __init__(self, base_conv, force_model=None, **kwargs):
        super().__init__()
        self.base_conv = base_conv.lower()
        registry = {"gine": HomoMoleculeGNN_GINE}
        for other in ("gatv2", "gin", "gps", "pna", "attentivefp", "heat"):
            registry[other] = _not_accelerated(other)
        build = force_model if force_model is not None else registry[self.base_conv]
        self.gnn_model = build(**kwargs)

    def forward(self, x, edge_index, ntypes, etypes, eattr=None, batch=None):
        return self.gnn_model(x, edge_index, ntypes, etypes, eattr=eattr, batch=batch)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("gnn_model"), name)


def _not_accelerated(name):
    def build(**kwargs):
        raise NotImplementedError(
            f"molecule encoder '{name}' is outside the MI355X hot path of this build (train_model.py:294 "
            "selects 'gine'); see DESIGN.md, 'Out of scope'")
    return build


class BaseMoleculeGNN(nn.Module):
    def __init__(self, in_channels, edge_dim, num_ntypes, num_etypes, ntype_emb_dim, etype_emb_dim,
                 num_convs=1, hidden_channels=None, out_channels=8, dropout_rate=0.2, activation="relu",
                 aggr="sum"):
        super().__init__()
        self.in_channels = in_channels
        self.edge_dim = edge_dim
        self.num_ntypes = num_ntypes
        self.num_etypes = num_etypes
        self.num_convs = num_convs
        self.hidden_channels = hidden_channels if hidden_channels is not None else out_channels
        self.out_channels = out_channels
        self.dropout_rate = dropout_rate
        self.aggr = aggr
        self.ntype_emb_dim = ntype_emb_dim
        self.etype_emb_dim = etype_emb_dim
        self._onehot_ntypes = ntype_emb_dim is None
        self._onehot_etypes = etype_emb_dim is None
        if self._onehot_ntypes:
            self.ntype_embedding = partial(nn.functional.one_hot, num_classes=num_ntypes)
            self.ntype_emb_dim = num_ntypes
        else:
            self.ntype_embedding = nn.Embedding(num_ntypes, ntype_emb_dim)
        if self._onehot_etypes:
            self.etype_embedding = partial(nn.functional.one_hot, num_classes=num_etypes)
            self.etype_emb_dim = num_etypes
        else:
            self.etype_embedding = nn.Embedding(num_etypes, etype_emb_dim)
        self.activation = _select_activation(activation)
        self.dropout = nn.Dropout(dropout_rate)

    def _embed_types_and_cat(self, x, eattr, ntypes, etypes):
        x = torch.cat([self.ntype_embedding(ntypes), x], dim=-1)
        eattr = torch.cat([self.etype_embedding(etypes), eattr], dim=-1)
        return x, eattr


class MLP(nn.Module):
    """Parameter container named like PyG's MLP: `lins` = [Linear(in, hid), Linear(hid, out)]."""

    def __init__(self, channels):
        super().__init__()
        self.lins = nn.ModuleList([nn.Linear(a, b) for a, b in zip(channels[:-1], channels[1:])])

    def __repr__(self):
        chans = [self.lins[0].in_features] + [l.out_features for l in self.lins]
        return f"MLP({', '.join(map(str, chans))})"


class GINEConv(nn.Module):
    """Parameter container named like PyG's GINEConv: `nn` (the MLP), `lin`
    (edge features -> node width) and `eps` (shape [1])."""

    def __init__(self, mlp, train_eps, edge_dim, in_channels, aggr="sum"):
        super().__init__()
        self.nn = mlp
        self.aggr = aggr
        eps0 = torch.zeros(1)
        if train_eps:
            self.eps = nn.Parameter(eps0)
        else:
            self.register_buffer("eps", eps0)
        self.lin = nn.Linear(edge_dim, in_channels)

    def __repr__(self):
        return f"GINEConv(nn={self.nn})"

    def kernel_weights(self, etype_embedding=None):
        """The seven tensors of the layer as the kernels take them.  `etype_embedding` (an nn.Embedding, when the model
        encodes bond types that way, molecule_gnn.py:118-122): the edge Linear sees [emb(type) | features]; its embedding
        part is linear in the table, so an EQUIVALENT one-hot weight [W[:, :D] @ table^T | W[:, D:]] is built here (one
        tiny matmul + cat, differentiable: autograd carries the kernels' gradient back to `lin.weight` and the table)."""
        l0, l1 = self.nn.lins
        we = self.lin.weight
        if etype_embedding is not None:
            d = etype_embedding.embedding_dim
            we = torch.cat([we[:, :d] @ etype_embedding.weight.t(), we[:, d:]], dim=1)
        return dict(eps=self.eps, we=we, be=self.lin.bias, w0=l0.weight, b0=l0.bias, w1=l1.weight, b1=l1.bias)


class HomoMoleculeGNN_GINE(BaseMoleculeGNN):
    """GINE encoder over atom graphs: N x 41 atom features -> N x 64 atom embeddings."""

    def __init__(self, act_first=False, gin_norm=None, gin_norm_kwargs=None, gin_trainable_eps=True, **kwargs):
        super().__init__(**kwargs)
        if gin_norm is not None:
            raise NotImplementedError("gin_norm is not compiled into the fused GINE kernel (the reference trains with "
                                      "gin_norm=None, train_model.py:300-312)")
        # act_first only reorders activation and normalisation inside PyG's MLP: without a norm it changes nothing
        if self.aggr not in ("sum", "add"):
            raise NotImplementedError("the fused GINE kernel implements aggr='sum'")
        self.act_first = act_first
        self.gin_norm = gin_norm
        self.gin_norm_kwargs = gin_norm_kwargs
        self.gin_trainable_eps = gin_trainable_eps
        if self.num_convs == 1:
            warnings.warn("The HomoMoleculeGNN_GINE model will not use the hidden_channels parameter "
                          "for a single convolution")
        widths = ([self.in_channels + self.ntype_emb_dim] + [self.hidden_channels] * (self.num_convs - 1)
                  + [self.out_channels])
        edge_w = self.edge_dim + self.etype_emb_dim
        self.conv_list = nn.ModuleList([
            GINEConv(MLP([a, b, b]), gin_trainable_eps, edge_w, a, aggr=self.aggr)
            for a, b in zip(widths[:-1], widths[1:])])
        self._widths = widths
        self._fused = False

    def fuse_parameters(self):
        """ONE trainable leaf (`arena`: per layer eps | nn.lins.0 | nn.lins.1 | lin, the order the kernels' gradient
        buffer has) instead of 7 per layer; see `gvp_hip.arena.FusedLeaf` and
        `VectorProteinGNN_LBAModel.fuse_parameters`.  One-hot type encoders and trainable eps only."""
        if self._fused:
            return self
        if not (self._onehot_ntypes and self._onehot_etypes) or not self.gin_trainable_eps:
            raise NotImplementedError("fuse_parameters: needs one-hot type encoders and gin_trainable_eps=True")
        from gvp_hip.arena import FusedLeaf
        keys = []
        for l in range(self.num_convs):
            keys += [f"conv_list.{l}.{k}" for k in ("eps", "nn.lins.0.weight", "nn.lins.0.bias", "nn.lins.1.weight",
                                                    "nn.lins.1.bias", "lin.weight", "lin.bias")]
        self._leaf = FusedLeaf(self, keys)
        self._fused = True
        return self

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if getattr(self, "_fused", False):
            self._leaf.seat()
        return out

    def forward(self, x, edge_index, ntypes, etypes, eattr=None, batch=None):
        slope = activation_slope(self.activation)
        if slope is None:
            raise NotImplementedError(f"activation {self.activation} is not compiled into the fused GINE kernel "
                                      "(ReLU / LeakyReLU / none are)")
        if not self._onehot_ntypes:
            # nn.Embedding atom types (molecule_gnn.py:112-116): the embedding row sits UNDER the message ReLU and the
            # (1 + eps) residual, so it cannot be folded into a weight; it is materialised (one gather + cat) and the
            # first layer runs as a plain [Na, D + F] layer.  The kernels are compiled for D + F = 52.
            if self.ntype_emb_dim + self.in_channels != 52:
                raise NotImplementedError("nn.Embedding atom types: the fused GINE kernel is compiled for ntype_emb_dim + "
                                          f"in_channels = 52, got {self.ntype_emb_dim} + {self.in_channels}")
            x = torch.cat([self.ntype_embedding(ntypes).to(x.dtype), x], dim=-1)
        if eattr is None:
            raise NotImplementedError("the GINE encoder needs edge features (eattr)")
        if not x.is_cuda:
            raise RuntimeError(f"x: caster-dta_amd runs on MI355X only (got a {x.device} tensor); there is no CPU path")
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or eattr.requires_grad
                                                  or any(p.requires_grad for p in self.parameters()))
        train_dropout = self.training and self.dropout_rate > 0 and self.num_convs > 1
        from gvp_hip import autograd_ops
        # ONE custom op (caster_gvp::gine_encoder) for the whole encoder: one launch per layer (+ the CSR build)
        return autograd_ops.gine_encoder(self, x, ntypes, eattr, etypes, edge_index, slope, train_dropout,
                                         save_state=bool(needs_grad or train_dropout))
