"""`models.joint_gnn`: the caller of the hot path, with the reference's API
(joint_gnn.py:15-288 JointGNN, :321-409 CrossAttentionModule, :411-451 stack).

The two encoders (`protein_gnn`, `molecule_gnn`) are the MI355X kernels.  The head
keeps the reference's modules and parameter names (checkpoints load strictly) but
runs on COMPACT rows: per-node Linear on [N, 128] / [Na, 128], the residue <-> atom
cross attention as ONE varlen kernel launch for both directions
(`caster_gvp::cross_attention`, csrc/attn_kernels.hip: no `to_dense_batch`
padding, no key-padding masks, no [B, heads, Rmax, Amax] score tensor), the
projections / feed-forward as library GEMMs on the compact rows, segment pooling
by `batch`, then the affinity MLP.  With `ptr` offsets in the input dicts (what
`forward_with_graphs` passes) the step has no data-dependent shape and no host
synchronisation, so it captures into a HIP graph.  The reference's dense
formulation (`to_dense_batch` + nn.MultiheadAttention) is kept for head shapes the
kernel is not compiled for (head_dim != 16, attention dropout > 0).
It does not import torch_geometric (`to_dense_batch` is restated below).

Multi-GPU: independent protein/drug pairs shard across ranks (one process per
GPU).  `enable_pair_parallel()` inserts ONE all-gather of the per-pair embedding
`cat[protein_embed, molecule_embed]` ([B_local, 512] fp32 -> [B, 512], RCCL over
xGMI) right before `pm_embed_lin` (joint_gnn.py:272-273); the affinity head then
runs replicated and every rank returns all B predictions.  Training adds ONE flat
all-reduce of the pre-gather parameter gradients (`reduce_pair_parallel_grads`).
"""
from functools import partial

import contextlib

import torch
import torch.nn as nn

from models.model_utils import _select_activation
from models.molecule_gnn import SelectableMoleculeModelWrapper as MoleculeGNN
from models.protein_gnn import SelectableProteinModelWrapper as ProteinGNN


def to_dense_batch(x, batch=None, batch_size=None):
    """[N, D] rows grouped by a sorted `batch` vector -> ([B, Lmax, D] zero padded,
    bool mask [B, Lmax]).  Same contract as torch_geometric.utils.to_dense_batch."""
    if batch is None:
        return x.unsqueeze(0), x.new_ones(1, x.shape[0], dtype=torch.bool)
    if batch_size is None:
        batch_size = int(batch.max()) + 1 if batch.numel() else 0
    counts = torch.bincount(batch, minlength=batch_size)
    lmax = int(counts.max()) if counts.numel() else 0
    start = torch.cumsum(counts, 0) - counts
    slot = batch * lmax + (torch.arange(batch.numel(), device=batch.device) - start[batch])
    dense = x.new_zeros(batch_size * lmax, x.shape[1])
    dense[slot] = x
    mask = torch.zeros(batch_size * lmax, dtype=torch.bool, device=x.device)
    mask[slot] = True
    return dense.view(batch_size, lmax, x.shape[1]), mask.view(batch_size, lmax)


class _BatchNorm(nn.Module):
    """Key-compatible stand-in for pyg.nn.BatchNorm (parameters live under `.module`)."""

    def __init__(self, channels, allow_single_element=True):
        super().__init__()
        self.module = nn.BatchNorm1d(channels)
        self.allow_single_element = allow_single_element

    def forward(self, x):
        if self.allow_single_element and x.shape[0] <= 1:
            return nn.functional.batch_norm(x, self.module.running_mean, self.module.running_var,
                                            self.module.weight, self.module.bias, False, 0.0, self.module.eps)
        return self.module(x)


_SIDE_STREAMS = {}


class _Lanes:
    """Two lanes for the head's row work (opt-in, `JointGNN.two_stream_head`): everything on the ATOM side -- its stack,
    LayerNorms, projections, feed-forward, pooling, molecule MLP: ~15 launches of a few microseconds each way on 2,560
    rows -- runs on a side stream beside the residue side (19,200 rows) instead of in between; the lanes meet at the
    attention core and at the pair concatenation.  Autograd runs every backward node on its forward's stream, so the
    backward overlaps the same way.  Tensors that cross lanes are recorded on the consuming stream (allocator safety)."""

    def __init__(self, on, device):
        self.on = bool(on)
        if self.on:
            self.main = torch.cuda.current_stream(device)
            key = (device.index, self.main.cuda_stream)
            if key not in _SIDE_STREAMS:
                _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
            self.side = _SIDE_STREAMS[key]

    def fork(self, *to_side):
        if self.on:
            self.side.wait_stream(self.main)
            for t in to_side:
                if t is not None:
                    t.record_stream(self.side)

    def join(self, *to_main):
        if self.on:
            self.main.wait_stream(self.side)
            for t in to_main:
                if t is not None:
                    t.record_stream(self.main)

    def atom(self):
        import contextlib
        return torch.cuda.stream(self.side) if self.on else contextlib.nullcontext()


class _JoinLanesAfterBackward(torch.autograd.Function):
    """Identity in the forward (placed where the lanes meet); its backward -- the FIRST node of the head's backward --
    queues an end-of-backward callback that makes the launch stream wait for the atom lane once more: the parameter
    gradients of the atom side are produced on that lane and consumed by nobody inside the graph, so without it a
    captured step would end with unjoined work (hipErrorStreamCaptureUnjoined) and an eager caller could read them early."""

    @staticmethod
    def forward(ctx, t, lanes):
        ctx.lanes = lanes
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        lanes = ctx.lanes
        torch.autograd.Variable._execution_engine.queue_callback(lambda: lanes.main.wait_stream(lanes.side))
        return g, None


class JointGNN(nn.Module):
    """Protein encoder + drug encoder + cross-attention + affinity head -> [B, 1]."""

    def __init__(self, protein_gnn_kwargs, molecule_gnn_kwargs, residue_lin_depth, atom_lin_depth,
                 n_attention_heads, attention_dropout, protein_lin_depth, molecule_lin_depth,
                 pairwise_embedding_dim, out_lin_depth, out_lin_factor=0.5, out_lin_norm_type=None,
                 activation="relu", dropout=0.0, element_pooling="mean", include_residual_stream=True,
                 residual_dim_ff_scale=2, num_cross_attn_layers=1, include_post_pool_layernorm=False):
        super().__init__()
        self.pairwise_embedding_dim = pairwise_embedding_dim
        self.n_attention_heads = n_attention_heads
        self.attention_dropout = attention_dropout
        self.element_pooling = element_pooling
        self.num_cross_attn_layers = num_cross_attn_layers
        self.include_residual_stream = include_residual_stream
        self.residual_dim_ff_scale = residual_dim_ff_scale
        self.include_post_pool_layernorm = include_post_pool_layernorm
        self.out_lin_factor = out_lin_factor
        self.out_lin_norm_type = out_lin_norm_type
        self.activation = _select_activation(activation)
        self.dropout = nn.Dropout(dropout)

        self.protein_gnn = ProteinGNN(**_tupled(protein_gnn_kwargs))
        self.molecule_gnn = MoleculeGNN(**molecule_gnn_kwargs)
        p_out, m_out = self.protein_gnn.out_channels, self.molecule_gnn.out_channels
        p_out = p_out[0] if isinstance(p_out, tuple) else p_out
        m_out = m_out[0] if isinstance(m_out, tuple) else m_out

        self.residue_lins, self.residue_norms, r_dim = self._make_lins_from_depth(residue_lin_depth, p_out)
        self.atom_lins, self.atom_norms, a_dim = self._make_lins_from_depth(atom_lin_depth, m_out)
        if num_cross_attn_layers > 0:
            block = partial(CrossAttentionModule, embed_dim_1=r_dim, embed_dim_2=a_dim,
                            n_attention_heads=n_attention_heads, attn_dropout=attention_dropout,
                            include_residual_stream=include_residual_stream,
                            dim_feedforward_scale=residual_dim_ff_scale, feedforward_dropout=dropout)
            self.cross_attn_module = StackedCrossAttentionModule(block, num_layers=num_cross_attn_layers)
        else:
            self.cross_attn_module = None
        if include_post_pool_layernorm:
            self.protein_post_pool_norm = nn.LayerNorm(r_dim)
            self.molecule_post_pool_norm = nn.LayerNorm(a_dim)
        self.protein_lins, self.protein_norms, p_dim = self._make_lins_from_depth(protein_lin_depth, r_dim)
        self.molecule_lins, self.molecule_norms, m_dim = self._make_lins_from_depth(molecule_lin_depth, a_dim)
        self.pm_embed_lin = nn.Linear(p_dim + m_dim, pairwise_embedding_dim)
        self.out_fc_layers, self.out_fc_norms, head_dim = self._make_lins_from_depth(
            out_lin_depth, pairwise_embedding_dim, scale_factor=out_lin_factor, include_norms=out_lin_norm_type)
        self.output_layer = nn.Linear(head_dim, 1)
        self._pair_group = None
        self._pair_parallel, self._pair_always = False, False
        self._pair_counts = None
        # nn.MultiheadAttention's head-averaged weights (second return value): "auto" = computed in eval mode (what
        # inference/evaluation.py:43-66 consumes), skipped in training (train_model.py:564 discards them);
        # "always" / "never" force it.  Dense [B, Rmax, Amax] / [B, Amax, Rmax] as in the reference.
        self.attention_weights = "auto"

    # ------------------------------------------------------------ multi-GPU
    def enable_pair_parallel(self, group=None, pair_counts=None, always_communicate=False):
        """Shard pairs over the ranks of `group`: ONE all-gather of the per-pair embeddings before the head.

        `pair_counts`: pairs held by every rank, as the sampler knows them (list of world_size ints).  None =
        every rank holds the same number of pairs (what a sharded sampler with drop_last / padding delivers);
        no count exchange and no host synchronisation happen inside the step either way.  Change it per step
        with `set_pair_counts` (e.g. a ragged last batch).

        Gradients: every rank evaluates the SAME full-batch loss on the gathered embeddings, so
          * parameters after the gather (pm_embed_lin, out_fc_layers, output_layer) get the complete gradient on
            every rank -- nothing to reduce;
          * parameters before it (both encoders, residue / atom lins, cross attention, protein / molecule lins)
            get the contribution of the LOCAL pairs only.  Call `reduce_pair_parallel_grads()` after
            `loss.backward()` and before `optimizer.step()`: one flat all-reduce(SUM) over RCCL.
        Do not wrap the model in DistributedDataParallel as well (it would average what must be summed).
        `always_communicate`: issue the collectives in a one-rank group too (they are skipped there by default);
        for rehearsing the N > 1 code path on a single GPU."""
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._pair_group, self._pair_parallel, self._pair_always = group, True, bool(always_communicate)
        self.set_pair_counts(pair_counts)
        return self

    def set_pair_counts(self, pair_counts):
        import torch.distributed as dist
        if pair_counts is not None:
            pair_counts = [int(c) for c in pair_counts]
            if len(pair_counts) != dist.get_world_size(self._pair_group):
                raise ValueError("pair_counts needs one entry per rank")
        self._pair_counts = pair_counts

    def _gather_pairs(self, pair):
        import torch.distributed as dist
        world = dist.get_world_size(self._pair_group)
        if world == 1 and not self._pair_always:
            return pair
        rank = dist.get_rank(self._pair_group)
        counts = self._pair_counts or [int(pair.shape[0])] * world
        if counts[rank] != pair.shape[0]:
            raise ValueError(f"rank {rank} holds {pair.shape[0]} pairs, pair_counts says {counts[rank]}")
        return _GatherPairs.apply(pair, counts, rank, self._pair_group)

    def pre_gather_parameters(self):
        """Parameters whose gradient is rank-local under pair parallelism (everything up to the gather)."""
        post = {id(p) for m in (self.pm_embed_lin, self.out_fc_layers, self.out_fc_norms, self.output_layer)
                for p in m.parameters()}
        return [p for p in self.parameters() if p.numel() and id(p) not in post]

    @torch.no_grad()
    def reduce_pair_parallel_grads(self):
        """All-reduce(SUM) of the pre-gather parameter gradients as ONE flat bucket (a few hundred KB: latency
        bound; one collective instead of ~90)."""
        import torch.distributed as dist
        if not self._pair_parallel or (dist.get_world_size(self._pair_group) == 1 and not self._pair_always):
            return
        ps = [p for p in self.pre_gather_parameters() if p.requires_grad]
        for p in ps:
            if p.grad is None:                    # e.g. a rank without pairs: contributes zeros
                p.grad = torch.zeros_like(p)
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self._pair_group)
        torch._foreach_copy_([p.grad for p in ps], [c.view_as(p) for c, p in zip(flat.split([p.numel() for p in ps]), ps)])

    def fuse_encoder_parameters(self):
        """Opt-in, eager-mode host time: the protein encoder's 74 parameter tensors and the drug encoder's 7 per layer
        become ONE trainable leaf each (`arena`), see `gvp_hip.arena.FusedLeaf`.  Checkpoints keep the reference's
        keys.  Call after `.to(device)` and before building the optimizer."""
        self.protein_gnn.gnn_model.fuse_parameters()
        self.molecule_gnn.gnn_model.fuse_parameters()
        return self

    # ------------------------------------------------------------ forward
    # "auto" (default): two lanes while the step is being CAPTURED into a HIP graph -- there the schedule is what the device
    # executes (whole-model step 1.56 -> 1.35 ms) -- one lane eagerly, where the step is host-bound and a second stream only
    # adds event calls; True / False force it
    two_stream_head = "auto"

    def _two_lanes(self):
        v = getattr(self, "two_stream_head", "auto")
        if v == "auto":
            return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        return bool(v)

    def forward_with_graphs(self, protein_graph, molecule_graph):
        return self.forward(*self._graphs_to_dicts(protein_graph, molecule_graph))

    @staticmethod
    def _graphs_to_dicts(protein_graph, molecule_graph):
        def as_dict(g):
            ei = g.edge_index if getattr(g, "edge_index", None) is not None else g.adj_t
            d = {"x": g.x, "edge_index": ei, "ntypes": g.node_type, "etypes": g.edge_type,
                 "eattr": g.edge_attr, "batch": g.batch}
            if getattr(g, "ptr", None) is not None:       # PyG Batch objects carry it: spares the head a host sync
                d["ptr"] = g.ptr
            return d
        return as_dict(protein_graph), as_dict(molecule_graph)

    def _stack(self, t, lins, norms, sites=None):
        from gvp_hip.head_ops import fast_linear      # F.linear; weight gradients of the per-row layers by the split-row kernel
        for lin, norm in zip(lins, norms):
            t = norm(fast_linear(t, lin.weight, lin.bias))
            # activation + dropout: one launch each way on the big compact-row stacks (gvp_hip.head_ops.DropSites)
            t = sites.act_dropout(t, self.activation, self.dropout) if sites is not None else self.dropout(self.activation(t))
        return t

    def _pool(self, dense, mask):
        m = mask.unsqueeze(-1)
        if self.element_pooling == "mean":
            return (dense * m).sum(dim=1) / mask.sum(dim=1, keepdim=True)
        if self.element_pooling == "sum":
            return (dense * m).sum(dim=1)
        if self.element_pooling == "max":
            return (dense - (~m) * 1.0e10).max(dim=1).values
        raise ValueError(self.element_pooling)

    def _pool_rows(self, rows, batch, ptr):
        """The same pooling on compact rows (graph b = rows ptr[b] .. ptr[b+1])."""
        B = ptr.numel() - 1
        if self.element_pooling in ("mean", "sum"):
            out = rows.new_zeros(B, rows.shape[1]).index_add_(0, batch, rows)
            if self.element_pooling == "mean":
                out = out / (ptr[1:] - ptr[:-1]).to(rows.dtype).unsqueeze(-1)
            return out
        if self.element_pooling == "max":
            idx = batch.unsqueeze(-1).expand_as(rows)
            return rows.new_full((B, rows.shape[1]), float("-inf")).scatter_reduce(0, idx, rows, "amax")
        raise ValueError(self.element_pooling)

    @staticmethod
    def _offsets(data, batch, rows):
        """(batch vector, ptr offsets) of a compact row array; `ptr` from the dict when given (no host sync)."""
        ptr = data.get("ptr", None)
        if batch is None:
            batch = torch.zeros(rows.shape[0], dtype=torch.long, device=rows.device)
            if ptr is None:
                ptr = torch.tensor([0, rows.shape[0]], dtype=torch.long, device=rows.device)
        if ptr is None:
            num = int(batch.max()) + 1 if batch.numel() else 0          # host sync; pass `ptr` to avoid it
            counts = torch.bincount(batch, minlength=num)
            ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
        return batch, ptr.to(torch.long)

    def forward(self, protein_graph_data={}, molecule_graph_data={}):
        pbatch = protein_graph_data.get("batch", None)
        mbatch = molecule_graph_data.get("batch", None)
        x0 = molecule_graph_data.get("x", None)
        lanes = None
        if torch.is_tensor(x0) and x0.is_cuda and not torch.compiler.is_compiling() and self._two_lanes():
            # opt-in two lanes: the drug encoder (and later the atom side of the head) on a side stream beside the protein
            # encoder -- forward and, because autograd runs a backward node on its forward's stream, backward
            lanes = _Lanes(True, x0.device)
            lanes.fork(*[t for t in molecule_graph_data.values() if torch.is_tensor(t)])
        residue = self.protein_gnn(**{k: v for k, v in protein_graph_data.items() if k != "ptr"})    # MI355X kernels
        with (lanes.atom() if lanes is not None else contextlib.nullcontext()):
            atom = self.molecule_gnn(**{k: v for k, v in molecule_graph_data.items() if k != "ptr"})     # MI355X kernels
        return self.head(residue, atom, protein_graph_data, molecule_graph_data, lanes)

    def head(self, residue, atom, protein_graph_data={}, molecule_graph_data={}, lanes=None):
        """Everything after the two encoders (joint_gnn.py:188-286): residue / atom stacks, cross attention, pooling,
        affinity head.  `residue` [N, D], `atom` [Na, D] are the encoders' outputs; the dicts supply batch / ptr."""
        pbatch = protein_graph_data.get("batch", None)
        mbatch = molecule_graph_data.get("batch", None)
        hdt = self.output_layer.weight.dtype
        if residue.dtype != hdt and not torch.is_autocast_enabled():   # bf16-storage encoders feeding an fp32 head
            residue = residue.to(hdt)
        if atom.dtype != hdt and not torch.is_autocast_enabled():
            with (lanes.atom() if lanes is not None else contextlib.nullcontext()):      # (the lane `atom` was produced on)
                atom = atom.to(hdt)
        from gvp_hip.head_ops import DropSites
        sites = DropSites(residue, self.training, self.dropout.p)     # fused dropout sites of this forward (eager fp32 training)
        varlen = self.cross_attn_module is None or self.cross_attn_module.varlen_supported(residue)
        if lanes is None or not varlen:
            if lanes is not None:
                lanes.join(atom)                  # dense fallback: everything back on the launch stream
            lanes = _Lanes(residue.is_cuda and varlen and not torch.compiler.is_compiling() and self._two_lanes(),
                           residue.device)
        lanes.fork(atom, sites.pair)
        residue = self._stack(residue, self.residue_lins, self.residue_norms, sites)
        with lanes.atom():
            atom = self._stack(atom, self.atom_lins, self.atom_norms, sites)
        attn = None
        if varlen:
            # compact rows end to end: varlen cross attention (one launch for both directions) + segment pooling
            pbatch, rptr = self._offsets(protein_graph_data, pbatch, residue)
            mbatch, aptr = self._offsets(molecule_graph_data, mbatch, atom)
            if rptr.numel() != aptr.numel():
                raise ValueError("protein and molecule batches hold different numbers of graphs")
            # `_offsets` may have CREATED aptr / mbatch on the launch stream (no 'ptr' / 'batch' from the caller); the atom
            # lane reads them below.  With a cross-attention module its own forks come later; without one nothing does.
            lanes.fork(aptr, mbatch)
            if self.cross_attn_module is not None:
                want = self.attention_weights == "always" or (self.attention_weights == "auto" and not self.training)
                residue, atom, attn = self.cross_attn_module.forward_varlen(residue, atom, rptr, aptr, want, sites, lanes)
            protein = self._pool_rows(residue, pbatch, rptr)
            with lanes.atom():
                molecule = self._pool_rows(atom, mbatch, aptr)
        else:
            # the reference's dense formulation (head shapes the varlen kernel is not compiled for)
            residue, rmask = to_dense_batch(residue, pbatch)
            atom, amask = to_dense_batch(atom, mbatch)
            residue, atom, attn = self.cross_attn_module(residue, atom, rmask, amask)
            protein, molecule = self._pool(residue, rmask), self._pool(atom, amask)
        if self.include_post_pool_layernorm:
            protein = self.protein_post_pool_norm(protein)
        protein = sites.act_dropout(protein, self.activation, self.dropout)
        protein = self._stack(protein, self.protein_lins, self.protein_norms, sites)
        with lanes.atom():
            if self.include_post_pool_layernorm:
                molecule = self.molecule_post_pool_norm(molecule)
            molecule = sites.act_dropout(molecule, self.activation, self.dropout)
            molecule = self._stack(molecule, self.molecule_lins, self.molecule_norms, sites)
        lanes.join(molecule)
        if lanes.on and torch.is_grad_enabled() and molecule.requires_grad:
            molecule = _JoinLanesAfterBackward.apply(molecule, lanes)
        pair = torch.cat([protein, molecule], dim=-1)
        if self._pair_parallel:
            pair = self._gather_pairs(pair)
        z = sites.act_dropout(self.pm_embed_lin(pair), self.activation, self.dropout)
        z = self._stack(z, self.out_fc_layers, self.out_fc_norms, sites)
        return self.output_layer(z), attn

    @staticmethod
    def _make_lins_from_depth(depth, in_dim, scale_factor=2, include_norms=None):
        """`depth` Linear layers, each scaling the width by `scale_factor` (truncated)."""
        norm = {"layer": nn.LayerNorm, "batch": _BatchNorm}.get(include_norms, nn.Identity)
        lins, norms, width = [], [], in_dim
        for _ in range(depth):
            nxt = int(width * scale_factor)
            lins.append(nn.Linear(width, nxt))
            norms.append(norm(nxt))
            width = nxt
        return nn.ModuleList(lins), nn.ModuleList(norms), width


class _GatherPairs(torch.autograd.Function):
    """All-gather of the per-pair embeddings ([B_r, D] per rank -> [B, D] on every rank, rank order) as ONE
    collective (`all_gather_into_tensor`, shards padded to the largest rank).  Backward: the head and its loss
    are replicated, so the gradient of the gathered tensor is already complete on every rank and the local
    gradient is its own slice -- no collective in the backward."""

    @staticmethod
    def forward(ctx, pair, counts, rank, group):
        import torch.distributed as dist
        world, width = len(counts), max(counts)
        padded = pair if pair.shape[0] == width else torch.cat(
            [pair, pair.new_zeros(width - pair.shape[0], pair.shape[1])])
        out = pair.new_empty(world * width, pair.shape[1])
        dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
        ctx.lo, ctx.n = sum(counts[:rank]), counts[rank]
        if all(c == width for c in counts):
            return out
        return torch.cat([out[r * width:r * width + c] for r, c in enumerate(counts)])

    @staticmethod
    def backward(ctx, g):
        return g[ctx.lo:ctx.lo + ctx.n], None, None, None


def _tupled(kwargs):
    """model_kwargs.json stores (s, v) dims as lists; the encoders expect tuples."""
    out = dict(kwargs)
    for k in ("in_channels", "edge_dim", "hidden_channels", "edge_hidden_channels", "out_channels"):
        if isinstance(out.get(k), list):
            out[k] = tuple(out[k])
    return out


class CrossAttentionModule(nn.Module):
    """Pre-norm bidirectional cross attention (residues <-> atoms) with optional
    residual feed-forward streams."""

    def __init__(self, embed_dim_1, embed_dim_2, n_attention_heads, attn_dropout, include_residual_stream=True,
                 dim_feedforward_scale=2, feedforward_dropout=0.2):
        super().__init__()
        self.include_residual_stream = include_residual_stream
        self.preattn_norm1 = nn.LayerNorm(embed_dim_1)
        self.preattn_norm2 = nn.LayerNorm(embed_dim_2)
        self.embed1_to_2 = nn.MultiheadAttention(embed_dim=embed_dim_1, kdim=embed_dim_2, vdim=embed_dim_2,
                                                 num_heads=n_attention_heads, dropout=attn_dropout,
                                                 batch_first=True)
        self.embed2_to_1 = nn.MultiheadAttention(embed_dim=embed_dim_2, kdim=embed_dim_1, vdim=embed_dim_1,
                                                 num_heads=n_attention_heads, dropout=attn_dropout,
                                                 batch_first=True)
        self.ff_norm1 = nn.LayerNorm(embed_dim_1)
        self.ff_norm2 = nn.LayerNorm(embed_dim_2)
        self.ff_dropout = nn.Dropout(feedforward_dropout)
        if include_residual_stream:
            def ff(d):
                return nn.Sequential(nn.Linear(d, d * dim_feedforward_scale), nn.ReLU(),
                                     nn.Dropout(feedforward_dropout), nn.Linear(d * dim_feedforward_scale, d))
            self.ff1, self.ff2 = ff(embed_dim_1), ff(embed_dim_2)

    # ------------------------------------------------------------ varlen path (MI355X kernel)
    def varlen_supported(self, rows):
        """The varlen kernel is compiled for head_dim 16 without attention dropout, on CUDA fp32 rows."""
        m1, m2 = self.embed1_to_2, self.embed2_to_1
        return (rows.is_cuda and m1.head_dim == 16 and m2.head_dim == 16 and m1.embed_dim == m2.embed_dim
                and m1.num_heads == m2.num_heads and not (self.training and (m1.dropout > 0 or m2.dropout > 0)))

    @staticmethod
    def _proj_parts(mha):
        """(wq, bq, wkv, bkv, wk, bk, wv, bv) of an nn.MultiheadAttention: the packed in_proj weight / bias SPLIT once (the
        backward of a split is ONE cat of its parts' gradients; the backward of every slice is a zero-filled full-size
        tensor + a copy + an add -- 12 slices made 12 fills and 12 copies per step).  wkv / bkv: K and V packed, for the
        one-GEMM projection (None when the weights are separate or there is no bias)."""
        E, b = mha.embed_dim, mha.in_proj_bias
        if mha._qkv_same_embed_dim and b is not None:
            wq, wkv = mha.in_proj_weight.split([E, 2 * E])
            bq, bkv = b.split([E, 2 * E])
            return wq, bq, wkv, bkv, None, None, None, None
        bq, bk, bv = (None, None, None) if b is None else b.split([E, E, E])
        if mha._qkv_same_embed_dim:
            wq, wk, wv = mha.in_proj_weight.split([E, E, E])
        else:
            wq, wk, wv = mha.q_proj_weight, mha.k_proj_weight, mha.v_proj_weight
        return wq, bq, None, None, wk, bk, wv, bv

    @staticmethod
    def _q(parts, x_q):
        from gvp_hip.head_ops import fast_linear
        return fast_linear(x_q, parts[0], parts[1])

    @staticmethod
    def _kv(parts, x_kv, E):
        """Key and value projections: ONE [*, 2E] GEMM when the weights are packed (K and V read the same rows and their
        weights are adjacent in in_proj_weight), halves made contiguous for the attention kernel."""
        from gvp_hip.head_ops import fast_linear
        if parts[2] is not None:
            k, v = fast_linear(x_kv, parts[2], parts[3]).split([E, E], dim=1)
            return k.contiguous(), v.contiguous()
        return fast_linear(x_kv, parts[4], parts[5]), fast_linear(x_kv, parts[6], parts[7])

    def forward_varlen(self, embed_1, embed_2, ptr1, ptr2, need_weights=False, sites=None, lanes=None):
        """`forward` on compact rows: embed_1 [N1, D] with graph offsets ptr1, embed_2 [N2, D] with ptr2; every
        graph b of side 1 attends to graph b of side 2 and vice versa (joint_gnn.py:376-398).  The softmax(QK^T)V
        core of both directions is one launch of caster_gvp::cross_attention; projections and feed-forward are
        GEMMs on the compact rows.  Returns (embed_1, embed_2, (w1, w2) or None)."""
        from gvp_hip import attention_ops  # noqa: F401  (registers the ops)
        from gvp_hip.head_ops import fast_layer_norm      # nn.LayerNorm on compact rows: one-pass kernels each way
        if lanes is None:
            lanes = _Lanes(False, embed_1.device)
        # side 1 (residue rows) on the launch stream, side 2 (atom rows) on the other lane: each side's norm and the
        # projections that read it (its own queries, the OTHER direction's keys / values)
        E = self.embed1_to_2.embed_dim
        p12, p21 = self._proj_parts(self.embed1_to_2), self._proj_parts(self.embed2_to_1)
        n1 = fast_layer_norm(embed_1, self.preattn_norm1)
        q1 = self._q(p12, n1)
        k2, v2 = self._kv(p21, n1, E)
        with lanes.atom():
            n2 = fast_layer_norm(embed_2, self.preattn_norm2)
            q2 = self._q(p21, n2)
            k1, v1 = self._kv(p12, n2, E)
        lanes.join(q2, k1, v1)
        heads = self.embed1_to_2.num_heads
        from gvp_hip import head_ops
        br = head_ops._bridge()
        if br is not None and hasattr(br, "head_cross_attention") and not torch.is_autocast_enabled("cuda"):
            o1, o2, lse1, lse2 = br.head_cross_attention(q1, k1, v1, q2, k2, v2, ptr1, ptr2, heads)     # eager: C++ autograd function
        else:
            o1, o2, lse1, lse2 = torch.ops.caster_gvp.cross_attention(q1, k1, v1, q2, k2, v2, ptr1, ptr2, heads)
        from gvp_hip.head_ops import fast_linear
        lanes.fork(o2)
        a1 = fast_linear(o1, self.embed1_to_2.out_proj.weight, self.embed1_to_2.out_proj.bias)
        with lanes.atom():
            a2 = fast_linear(o2, self.embed2_to_1.out_proj.weight, self.embed2_to_1.out_proj.bias)
        weights = None
        if need_weights:
            l1 = int((ptr1[1:] - ptr1[:-1]).max()) if ptr1.numel() > 1 else 0      # inference only: host sync
            l2 = int((ptr2[1:] - ptr2[:-1]).max()) if ptr2.numel() > 1 else 0
            weights = tuple(torch.ops.caster_gvp.cross_attention_weights(q1.detach(), k1.detach(), lse1.detach(),
                                                                         q2.detach(), k2.detach(), lse2.detach(),
                                                                         ptr1, ptr2, heads, l1, l2))
        if self.include_residual_stream:
            if sites is None:
                from gvp_hip.head_ops import DropSites
                sites = DropSites(embed_1, False, 0.0)          # inactive: the stock ops
            def ff(seq, t):          # nn.Sequential(Linear, ReLU, Dropout, Linear) with the row-wise Linear layers on fast_linear
                h = sites.act_dropout(fast_linear(t, seq[0].weight, seq[0].bias), seq[1], seq[2])
                return fast_linear(h, seq[3].weight, seq[3].bias)
            embed_1 = sites.dropout_add(embed_1, a1, self.ff_dropout)
            embed_1 = sites.dropout_add(embed_1, ff(self.ff1, fast_layer_norm(embed_1, self.ff_norm1)), self.ff_dropout)
            with lanes.atom():
                embed_2 = sites.dropout_add(embed_2, a2, self.ff_dropout)
                embed_2 = sites.dropout_add(embed_2, ff(self.ff2, fast_layer_norm(embed_2, self.ff_norm2)), self.ff_dropout)
        else:
            embed_1, embed_2 = a1, a2
        return embed_1, embed_2, weights          # (embed_2 stays on the atom lane: the caller joins)

    def forward(self, embed_1, embed_2, mask1, mask2, return_weights=True):
        n1, n2 = self.preattn_norm1(embed_1), self.preattn_norm2(embed_2)
        a1, w1 = self.embed1_to_2(n1, n2, n2, key_padding_mask=~mask2)
        a2, w2 = self.embed2_to_1(n2, n1, n1, key_padding_mask=~mask1)
        if self.include_residual_stream:
            embed_1 = embed_1 + self.ff_dropout(a1)
            embed_1 = embed_1 + self.ff_dropout(self.ff1(self.ff_norm1(embed_1)))
            embed_2 = embed_2 + self.ff_dropout(a2)
            embed_2 = embed_2 + self.ff_dropout(self.ff2(self.ff_norm2(embed_2)))
        else:
            embed_1, embed_2 = a1, a2
        return (embed_1, embed_2, (w1, w2)) if return_weights else (embed_1, embed_2)


class StackedCrossAttentionModule(nn.Module):
    def __init__(self, cross_attn_base, num_layers):
        super().__init__()
        self.cross_attn_layers = nn.ModuleList([cross_attn_base() for _ in range(num_layers)])

    def varlen_supported(self, rows):
        return all(layer.varlen_supported(rows) for layer in self.cross_attn_layers)

    def forward_varlen(self, embed_1, embed_2, ptr1, ptr2, need_weights=False, sites=None, lanes=None):
        weights = []
        for layer in self.cross_attn_layers:
            embed_1, embed_2, w = layer.forward_varlen(embed_1, embed_2, ptr1, ptr2, need_weights, sites, lanes)
            weights.append(w)
        return embed_1, embed_2, (weights if need_weights else None)

    def forward(self, embed_1, embed_2, mask1, mask2, return_weights=True):
        weights = []
        for layer in self.cross_attn_layers:
            embed_1, embed_2, w = layer(embed_1, embed_2, mask1, mask2, return_weights=True)
            weights.append(w)
        return (embed_1, embed_2, weights) if return_weights else (embed_1, embed_2)
