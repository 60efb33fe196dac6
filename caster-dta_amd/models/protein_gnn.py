"""`models.protein_gnn`: the residue-graph encoder behind the reference's module
API (protein_gnn.py:14-82 wrapper, :86-152 base, :289-388 LBA model).

`VectorProteinGNN_LBAModel` owns exactly the reference's parameters (same names
and shapes) but keeps them as views into one fp32 arena and runs its forward as
1 + 2*num_convs launches of libcaster_gvp.so on MI355X:

    node embed            gvp_node (GVP + LayerNorm)                 per residue
    conv l                gvp_edge + 3-GVP message + segmented sum   per edge / target
    node update l         residual+LN, 2-GVP feed-forward, residual+LN
                          (+ gvp_norm_before_scalar + gvp_to_scalar on the last layer)

There is no eager/CPU fallback: CPU tensors, a missing library or dimensions the
kernels were not compiled for raise.
"""
from functools import partial

import torch
import torch.nn as nn

import models.gvp_layers as gvp
from gvp_hip import ops
from gvp_hip.arena import ParamArena, lba_param_keys
from models.model_utils import _select_activation

_VECTOR_MODELS = ("lbamodel", "pocketminer", "cpdmodel")


class SelectableProteinModelWrapper(nn.Module):
    """Picks the protein encoder by `base_conv`; forwards calls and unknown
    attributes (e.g. `.out_channels`, `.num_ntypes`) to the wrapped model."""

    def __init__(self, in_channels, edge_dim, base_conv, **kwargs):
        super().__init__()
        if type(in_channels) is not type(edge_dim):
            raise ValueError("in_channels and edge_dim must be the same type - either both are ints to "
                             "represent scalars or both are tuples to represent (scalar, vector)")
        self.base_conv = base_conv
        self.is_scalar_data = isinstance(in_channels, int)
        if self.is_scalar_data and base_conv in _VECTOR_MODELS:
            raise ValueError(f"Cannot use a vector model {base_conv} with scalar input data {in_channels} "
                             "(either define the input as (n, 0) or include vector data)")
        if not self.is_scalar_data and base_conv not in _VECTOR_MODELS:
            raise ValueError(f"Cannot use a scalar model {base_conv} with vector input data {in_channels} "
                             "(either define the input as n or exclude vector data)")
        registry = {
            "lbamodel": VectorProteinGNN_LBAModel,
            "pocketminer": VectorProteinGNN_PocketMiner,
            "cpdmodel": VectorProteinGNN_CPDModel,
            "gatv2": _not_accelerated("HomoScalarProteinGNN_GATv2"),
            "heat": _not_accelerated("HeteroScalarProteinGNN_HEAT"),
        }
        self.gnn_model = registry[base_conv](in_channels=in_channels, edge_dim=edge_dim, **kwargs)

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
            f"{name} is outside the MI355X hot path of this build (the GVP encoders 'lbamodel' -- the default of "
            "train_model.py:276, hand-written kernels -- 'pocketminer' and 'cpdmodel' are implemented); see DESIGN.md, "
            "'Out of scope'")
    return build


class BaseProteinGNN(nn.Module):
    """Shared constructor state: channel bookkeeping and the type encoders
    (one-hot by default, nn.Embedding when an embedding width is given)."""

    def __init__(self, in_channels, edge_dim, num_ntypes, num_etypes, ntype_emb_dim, etype_emb_dim,
                 num_convs=1, hidden_channels=None, out_channels=8, dropout_rate=0.2, activation="relu"):
        super().__init__()
        self.in_channels = in_channels
        self.edge_dim = edge_dim
        self.num_ntypes = num_ntypes
        self.num_etypes = num_etypes
        self.num_convs = num_convs
        self.hidden_channels = hidden_channels if hidden_channels is not None else out_channels
        self.out_channels = out_channels
        self.dropout_rate = dropout_rate
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
        """Type encodings go IN FRONT of the features (column order of ws.weight)."""
        x = torch.cat([self.ntype_embedding(ntypes), x], dim=-1)
        eattr = torch.cat([self.etype_embedding(etypes), eattr], dim=-1)
        return x, eattr


class VectorProteinGNN_LBAModel(BaseProteinGNN):
    """GVP-GNN residue encoder (LBA-style): (N x 17, N x 3 x 3) residue features
    over a radius / kNN graph -> N x 64 residue embeddings."""

    def __init__(self, edge_hidden_channels, aggr="mean", **kwargs):
        super().__init__(**kwargs)
        self.edge_hidden_channels = edge_hidden_channels
        self.aggr = aggr
        if isinstance(self.hidden_channels, int):
            self.hidden_channels = (self.hidden_channels, 0)
        if isinstance(self.out_channels, int):
            self.out_channels = (self.out_channels, 0)
        self.in_channels = tuple(self.in_channels)
        self.edge_dim = tuple(self.edge_dim)
        self.hidden_channels = tuple(self.hidden_channels)
        self.edge_hidden_channels = tuple(edge_hidden_channels)
        node_in = (self.in_channels[0] + self.ntype_emb_dim, self.in_channels[1])
        edge_in = (self.edge_dim[0] + self.etype_emb_dim, self.edge_dim[1])
        plain = dict(activations=(None, None), vector_gate=True)
        self.gvp_node = nn.Sequential(gvp.GVP(node_in, self.hidden_channels, **plain),
                                      gvp.LayerNorm(self.hidden_channels))
        self.gvp_edge = nn.Sequential(gvp.GVP(edge_in, self.edge_hidden_channels, **plain),
                                      gvp.LayerNorm(self.edge_hidden_channels))
        self.gvp_relu = nn.ReLU()
        self.conv_list = nn.ModuleList([
            gvp.GVPConvLayer(self.hidden_channels, self.edge_hidden_channels, drop_rate=self.dropout_rate,
                             activations=(self.gvp_relu, None), vector_gate=True, aggr=aggr)
            for _ in range(self.num_convs)])
        self.gvp_norm_before_scalar = gvp.LayerNorm(self.hidden_channels)
        self.gvp_to_scalar = gvp.GVP(self.hidden_channels, self.out_channels,
                                     activations=(self.gvp_relu, None), vector_gate=True)
        named = dict(self.named_parameters())
        self._arena = ParamArena([named[k] for k in lba_param_keys(self.num_convs)])
        self._hip_cfg = None
        self._fused = False

    # ------------------------------------------------------------------ one-leaf mode (opt-in)
    def fuse_parameters(self):
        """Make the parameter arena itself the model's ONE trainable leaf, `arena` [15,117 floats for CASTER-DTA(2,2)].

        Why: in eager mode PyTorch's autograd engine spends 3-4 us per parameter leaf after the backward kernels have
        been issued (one AccumulateGrad node each); this encoder has 74 leaves, i.e. ~0.3 ms of host time per step --
        more than the whole step takes on the GPU.  After this call the backward pass hands the engine ONE gradient
        (the gradient arena the kernels write anyway), and an optimizer built over `model.parameters()` updates the
        arena in place (element-wise optimizers -- SGD, Adam, AdamW -- compute exactly what they compute per tensor).
        Checkpoints keep the reference's keys; `named_parameters()` lists `arena` (gvp_hip.arena.FusedLeaf).
        One-hot type encoders only.  Call it after the model is on its device and BEFORE building the optimizer."""
        if self._fused:
            return self
        if not (self._onehot_ntypes and self._onehot_etypes):
            raise NotImplementedError("fuse_parameters: nn.Embedding type encoders are folded per call (op_params) and "
                                      "keep their own leaves")
        from gvp_hip.arena import FusedLeaf
        self._leaf = FusedLeaf(self, lba_param_keys(self.num_convs))
        self._fused = True
        return self

    # ------------------------------------------------------------------ HIP path
    def _hip_config(self):
        if self._hip_cfg is None:
            if self.out_channels[1] != 0:
                raise NotImplementedError("the fused head produces scalars only (out_channels = (n, 0))")
            dims = ops.make_dims(node_in_s=self.in_channels[0], node_in_v=self.in_channels[1],
                                 edge_in_s=self.edge_dim[0], edge_in_v=self.edge_dim[1],
                                 hidden_s=self.hidden_channels[0], hidden_v=self.hidden_channels[1],
                                 edge_hidden_s=self.edge_hidden_channels[0],
                                 edge_hidden_v=self.edge_hidden_channels[1], out_s=self.out_channels[0])
            layout = ops.lba_layout(dims, self.num_ntypes, self.num_etypes, self.num_convs)
            self._hip_cfg = (dims, layout)
        return self._hip_cfg

    def op_params(self):
        """The arena-ordered parameter list the encoder op takes.  With the default one-hot type encoders these are the
        nn.Parameters themselves (views into the arena).  With nn.Embedding type encoders (`ntype_emb_dim` /
        `etype_emb_dim`, protein_gnn.py:123-133) the embedding is FOLDED into the first GVP's `ws`: the kernels add the
        column `ws.weight[:, type]` of a one-hot layout, and `ws.weight[:, :D] @ embedding.weight[type]` is exactly
        what Linear(cat[embedding(type), s, |v|]) contributes -- so an equivalent [so, num_types + si + h] weight is
        built here (one tiny matmul + cat per call, differentiable: autograd carries the kernels' gradient of the
        equivalent weight back to `ws.weight` and to the embedding table)."""
        if self._fused:
            return [self.arena]
        params = list(self._arena.params)
        if self._onehot_ntypes and self._onehot_etypes:
            return params
        keys = [k for k in lba_param_keys(self.num_convs)]
        named = dict(self.named_parameters())
        live = [k for k in keys if named[k].numel() > 0]
        for key, emb, onehot in (("gvp_node.0.ws.weight", self.ntype_embedding, self._onehot_ntypes),
                                 ("gvp_edge.0.ws.weight", self.etype_embedding, self._onehot_etypes)):
            if onehot:
                continue
            w = named[key]
            d = emb.embedding_dim
            params[live.index(key)] = torch.cat([w[:, :d] @ emb.weight.t(), w[:, d:]], dim=1)
        return params

    def _arena_buffer(self):
        """Re-seat the parameters as views into one flat buffer if something moved them (eager only; the custom
        op also accepts parameters that are NOT arena views -- it then concatenates them, one extra launch)."""
        dims, layout = self._hip_config()
        if self._arena.total != layout.total:
            raise RuntimeError(f"parameter arena has {self._arena.total} floats, kernels expect {layout.total}")
        return self._arena.buffer()

    def _apply(self, fn, *args, **kwargs):
        """.to() / .cuda() / .float() re-materialise every parameter: rebuild the arena right away, so that a
        model handed to torch.compile (which never runs the eager bookkeeping in forward) is already zero-copy."""
        out = super()._apply(fn, *args, **kwargs)
        if self._fused:
            self._leaf.seat()           # the arena moved / was re-materialised: re-point the per-module views
            return out
        p0 = self._arena.params[0]
        if p0.is_cuda and p0.dtype == torch.float32:
            self._arena.rebuild()
        return out

    def forward(self, x, edge_index, ntypes, etypes, eattr=None, batch=None):
        x_s, x_v = x
        if eattr is None:
            raise NotImplementedError("the LBA encoder needs edge features (eattr)")
        e_s, e_v = eattr
        if self.aggr not in ("sum", "add", "mean"):
            raise ValueError(f"unsupported aggregation {self.aggr!r}")
        if not x_s.is_cuda:
            raise RuntimeError(f"x: caster-dta_amd runs on MI355X only (got a {x_s.device} tensor); there is no CPU path")
        p0 = self.arena if self._fused else self._arena.params[0]
        if p0.dtype != torch.float32:
            raise TypeError("the MI355X kernels are fp32; call .float() on the model")
        if self.out_channels[1] != 0:
            raise NotImplementedError("the fused head produces scalars only (out_channels = (n, 0))")
        from gvp_hip import autograd_ops
        if not self._fused and self._onehot_ntypes and self._onehot_etypes and not torch.compiler.is_compiling() \
                and autograd_ops._eager_bridge() is None:       # (the C++ fast path checks the arena itself)
            self._arena_buffer()
        leaves = [self.arena] if self._fused else self._arena.params
        needs_grad = torch.is_grad_enabled() and (
            any(p.requires_grad for p in leaves) or any(t.requires_grad for t in (x_s, x_v, e_s, e_v)))
        train_dropout = self.training and self.dropout_rate > 0
        # ONE host call (C++ autograd node in eager mode, the caster_gvp::lba_encoder custom op under torch.compile)
        # for the whole encoder pass: 1 + num_convs launches (+ the CSR build)
        return autograd_ops.lba_encoder(self, x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, train_dropout,
                                        save_state=bool(needs_grad or train_dropout))


# ---------------------------------------------------------------------------------------------------------------------
# The reference's two other GVP stacks (protein_gnn.py:392-516 PocketMiner-style, :518-608 CPD-style with an
# autoregressive decoder).  train_model.py never selects them (it hard-codes 'lbamodel', :276) and their channel widths
# are free constructor arguments, so they are built from this package's `models.gvp_layers` modules with the
# reference's parameter names / shapes (strict checkpoint loading).  On MI355X every `GVPConvLayer` in them whose dims
# are the compiled ones -- node (16, 4), edge (<= 32, 1) -- runs on the tile kernels in its own layer kind (un-gated
# `(None, None)` for PocketMiner, `(relu, sigmoid)` for CPD; the autoregressive decoder as two masked conv passes),
# forward and backward (gvp_hip/conv_layer_ops.py); the projection GVPs / LayerNorms around them, and layers at other
# widths, are tensor ops on whatever device the tensors live on.  Pinned against outputs and gradients of the
# reference's own classes (tests/golden/gvp_stacks.npz; CPU and GPU tests in tests/test_gvp_stacks.py).
class VectorProteinGNN_PocketMiner(BaseProteinGNN):
    """Structural projection GVPs -> type embedding cat -> LayerNorm + GVP on nodes and edges -> `num_convs`
    GVPConvLayers (activations (None, None), aggr 'mean', no gate) -> LayerNorm + GVP -> per-residue scalars."""

    def __init__(self, edge_hidden_channels, initial_node_project_channels, initial_edge_project_channels, **kwargs):
        super().__init__(**kwargs)
        self.edge_hidden_channels = edge_hidden_channels
        if isinstance(self.hidden_channels, int):
            self.hidden_channels = (self.hidden_channels, 0)
        if isinstance(self.out_channels, int):
            self.out_channels = (self.out_channels, 0)
        plain = dict(activations=(None, None))
        if initial_node_project_channels is None:
            self.gvp_node_structural_proj = nn.Identity()
            initial_node_project_channels = self.in_channels
        else:
            self.gvp_node_structural_proj = nn.Sequential(
                gvp.GVP(self.in_channels, initial_node_project_channels, **plain),
                gvp.LayerNorm(initial_node_project_channels))
        if initial_edge_project_channels is None:
            self.gvp_edge_structural_proj = nn.Identity()
            initial_edge_project_channels = self.edge_dim
        else:
            self.gvp_edge_structural_proj = nn.Sequential(
                gvp.GVP(self.edge_dim, initial_edge_project_channels, **plain),
                gvp.LayerNorm(initial_edge_project_channels))
        self.initial_node_project_channels = initial_node_project_channels
        self.initial_edge_project_channels = initial_edge_project_channels
        node_in = (initial_node_project_channels[0] + self.ntype_emb_dim, initial_node_project_channels[1])
        edge_in = (initial_edge_project_channels[0] + self.etype_emb_dim, initial_edge_project_channels[1])
        self.gvp_node = nn.Sequential(gvp.LayerNorm(node_in), gvp.GVP(node_in, self.hidden_channels, **plain))
        self.gvp_edge = nn.Sequential(gvp.LayerNorm(edge_in), gvp.GVP(edge_in, self.edge_hidden_channels, **plain))
        self.conv_list = nn.ModuleList([
            gvp.GVPConvLayer(self.hidden_channels, self.edge_hidden_channels, drop_rate=self.dropout_rate, **plain)
            for _ in range(self.num_convs)])
        self.gvp_norm_before_scalar = gvp.LayerNorm(self.hidden_channels)
        self.gvp_to_scalar = gvp.GVP(self.hidden_channels, self.out_channels, **plain)

    def forward(self, x, edge_index, ntypes, etypes, eattr=None, batch=None):
        x_s, x_v = x
        if eattr is None:
            eattr = (x_s.new_zeros(edge_index.shape[1], 0), x_s.new_zeros(edge_index.shape[1], 0, 3))
        x_s, x_v = self.gvp_node_structural_proj((x_s, x_v))
        e_s, e_v = self.gvp_edge_structural_proj(tuple(eattr))
        x_s, e_s = self._embed_types_and_cat(x_s, e_s, ntypes, etypes)
        h = self.gvp_node((x_s, x_v))
        e = self.gvp_edge((e_s, e_v))
        for conv in self.conv_list:
            h = conv(h, edge_index, edge_attr=e)
        return self.gvp_to_scalar(self.gvp_norm_before_scalar(h))


class VectorProteinGNN_CPDModel(BaseProteinGNN):
    """Encoder GVPConvLayers, then decoder GVPConvLayers that see the residue identity of EARLIER residues only
    (edge features extended with the source's type encoding, zeroed where src >= dst) and read the encoder's
    embeddings for messages with src >= dst (`autoregressive_x`, gvp_layers.py:382-398)."""

    def __init__(self, edge_hidden_channels, **kwargs):
        super().__init__(**kwargs)
        self.edge_hidden_channels = edge_hidden_channels
        if isinstance(self.hidden_channels, int):
            self.hidden_channels = (self.hidden_channels, 0)
        if isinstance(self.out_channels, int):
            self.out_channels = (self.out_channels, 0)
        plain = dict(activations=(None, None))
        edge_in = (self.edge_dim[0] + self.etype_emb_dim, self.edge_dim[1])
        self.W_v = nn.Sequential(gvp.GVP(self.in_channels, self.hidden_channels, **plain),
                                 gvp.LayerNorm(self.hidden_channels))
        self.W_e = nn.Sequential(gvp.GVP(edge_in, self.edge_hidden_channels, **plain),
                                 gvp.LayerNorm(self.edge_hidden_channels))
        self.encoder_layers = nn.ModuleList(
            gvp.GVPConvLayer(self.hidden_channels, self.edge_hidden_channels, drop_rate=self.dropout_rate)
            for _ in range(self.num_convs))
        dec_edge = (self.edge_hidden_channels[0] + self.ntype_emb_dim, self.edge_hidden_channels[1])
        self.decoder_layers = nn.ModuleList(
            gvp.GVPConvLayer(self.hidden_channels, dec_edge, drop_rate=self.dropout_rate, autoregressive=True)
            for _ in range(self.num_convs))
        self.W_out = gvp.GVP(self.hidden_channels, self.out_channels, **plain)

    def forward(self, x, edge_index, ntypes, etypes, eattr=None, batch=None):
        if eattr is None:
            eattr = (x[0].new_zeros(edge_index.shape[1], 0), x[0].new_zeros(edge_index.shape[1], 0, 3))
        e_s = torch.cat([self.etype_embedding(etypes), eattr[0]], dim=-1)
        h = self.W_v(x)
        e = self.W_e((e_s, eattr[1]))
        for layer in self.encoder_layers:
            h = layer(h, edge_index, e)
        enc = h
        h_s = self.ntype_embedding(ntypes)[edge_index[0]]
        h_s = torch.where((edge_index[0] >= edge_index[1]).unsqueeze(-1), torch.zeros_like(h_s), h_s)
        e = (torch.cat([e[0], h_s.to(e[0].dtype)], dim=-1), e[1])
        for layer in self.decoder_layers:
            h = layer(h, edge_index, e, autoregressive_x=enc)
        return self.W_out(h)
