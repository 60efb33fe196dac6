"""Host-side launch plumbing for the HIP kernels: tensor checks, workspace
allocation (through PyTorch's caching allocator) and calls into the C ABI on
PyTorch's current HIP stream.  No arithmetic happens here.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import torch

from . import _lib
from ._lib import Dims, GineW, Layout, Rng

CASTER_DIMS = dict(node_in_s=17, node_in_v=3, edge_in_s=32, edge_in_v=1, hidden_s=16, hidden_v=4,
                   edge_hidden_s=32, edge_hidden_v=1, out_s=64, storage=0, layer_kind=0)
LAYER_GATED, LAYER_GVPDEF, LAYER_LINEAR = 0, 1, 2   # cgvp_dims.layer_kind (include/caster_gvp.h, "LAYER KIND")
F32, BF16 = 0, 1   # cgvp_dims.storage: element type of the activation buffers ("bf16 storage / fp32 accumulate")
ROW = 28  # merged node row: 16 scalars + 4x3 vector channels
EROW = 36  # stored edge embedding row (CGVP_EDGE_ROW): 32 scalars + 1x3 vector + pad, sorted-edge order


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else C.c_void_p(0)


def _f32(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: caster-dta_amd runs on MI355X only (got a {t.device} tensor); "
                           "there is no CPU path")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def _act(t, name, dtype):
    """An ACTIVATION buffer (features, node rows, edge-embedding store): contiguous, 16-B aligned, of the pass's
    storage dtype (torch.float32 or torch.bfloat16 -- cast here if the caller mixed them)."""
    if not t.is_cuda:
        raise RuntimeError(f"{name}: caster-dta_amd runs on MI355X only (got a {t.device} tensor); "
                           "there is no CPU path")
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"{name}: expected float32 or bfloat16, got {t.dtype}")
    if t.dtype != dtype:
        t = t.to(dtype)
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def storage_dtype(*tensors):
    """bf16 storage when the node features arrive in bf16, fp32 otherwise."""
    return torch.bfloat16 if tensors[0].dtype == torch.bfloat16 else torch.float32


def _i64(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: caster-dta_amd runs on MI355X only (got a {t.device} tensor)")
    if t.dtype != torch.int64:
        t = t.long()
    return t.contiguous()


def make_rng(seed, p, stream):
    """cgvp_rng for in-kernel dropout: `seed` = int64[2] CUDA tensor {seed, offset}; None -> NULL (no dropout)."""
    if seed is None or seed.numel() == 0 or p <= 0:
        return None
    return Rng(seed.data_ptr(), float(p), int(stream))


def _rng_ref(rng):
    return C.byref(rng) if rng is not None else None


def dropout_masks(seed, p, stream, num_masks, num_nodes, width):
    """The factors the kernels apply for (seed, p, stream .. stream + num_masks - 1): [num_masks, N, width]
    (width 20 = protein row [16 scalar | 4 vector-channel]; otherwise the GINE row)."""
    out = torch.empty(num_masks, num_nodes, width, dtype=torch.float32, device=seed.device)
    rng = make_rng(seed, p, stream)
    with torch.cuda.device(seed.device):
        _lib.check(_lib.lib().cgvp_dropout_masks(C.byref(rng), num_masks, num_nodes, width, _ptr(out), _stream()),
                   "cgvp_dropout_masks")
    return out


def make_dims(**kw):
    d = dict(CASTER_DIMS)
    d.update(kw)
    return Dims(**d)


def lba_layout(dims, num_ntypes, num_etypes, num_convs):
    out = Layout()
    _lib.check(_lib.lib().cgvp_lba_layout(C.byref(dims), num_ntypes, num_etypes, num_convs, C.byref(out)),
               "cgvp_lba_layout")
    return out


@dataclass
class Csr:
    """Destination-sorted CSR of a batched graph (int32 tables on the GPU)."""
    rowptr: torch.Tensor
    eperm: torch.Tensor
    esrc: torch.Tensor
    edst: torch.Tensor
    num_nodes: int
    num_edges: int
    table_rows: int = None      # set by CsrStore.collate(table=True): eperm indexes a resident feature table of this many rows


def build_csr(edge_index, num_nodes, counted=None):
    """`counted`: the counters buffer a cgvp_lba_pass_begin launch on this stream has already filled with the per-target
    edge counts of this very edge_index (csr_counters(...)): the build then skips its count launch."""
    try:
        ei = _i64(edge_index, "edge_index")
        if ei.dim() != 2 or ei.shape[0] != 2:
            raise ValueError(f"edge_index must be [2, E], got {tuple(ei.shape)}")
    except Exception:
        if counted is not None:
            counted.zero_()        # a pass_begin launch has already counted into them
        raise
    E = int(ei.shape[1])
    dev = ei.device
    i32 = dict(dtype=torch.int32, device=dev)
    rowptr = torch.empty(num_nodes + 1, **i32)
    eperm = torch.empty(max(E, 1), **i32)
    esrc = torch.empty(max(E, 1), **i32)
    edst = torch.empty(max(E, 1), **i32)
    ids = torch.empty(max(E, 1), **i32)
    with torch.cuda.device(dev):
        work = counted if counted is not None else csr_counters(dev, num_nodes)
        rc = _lib.lib().cgvp_csr_from_coo(_ptr(ei), num_nodes, E, _ptr(rowptr), _ptr(eperm), _ptr(esrc),
                                          _ptr(edst), _ptr(work), 2 if counted is not None else 1, _ptr(ids), _stream())
    if rc != 0:
        work.zero_()               # the persistent counters must not stay half-used (the next build assumes zeros)
    _lib.check(rc, "cgvp_csr_from_coo")
    return Csr(rowptr, eperm, esrc, edst, num_nodes, E)


def csr_counters(dev, num_nodes):
    """The zeroed per-target counters of a CSR build on the current stream (see _zeroed_counters)."""
    return _zeroed_counters(dev, (num_nodes + 1 + 63) // 64 * 64)


_COUNTERS = {}      # (device index, stream handle) -> [int32 counters, newest last]; zero between calls (every build leaves them so)


def _zeroed_counters(dev, n):
    """The CSR build's per-target counters: one persistent buffer per (device, stream), zero-filled ONCE when it is
    created or grown -- every build leaves the counters it used zeroed again, so a step needs no fill launch.  Keyed by
    stream because builds on different streams (protein / drug) run concurrently.
    LIFETIME: a buffer handed out once is never freed -- a HIP graph captured with it keeps counting into it on every
    replay, whatever the eager code does afterwards.  Growing (a larger batch arrives) appends a new generation and
    keeps the old ones alive; sizes double, so the dead weight is bounded by the live size."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    gens = _COUNTERS.setdefault(key, [])
    if not gens or gens[-1].numel() < n:
        # (created inside a stream capture -- no warm-up ran on this stream -- the fill is recorded into the graph and
        # re-zeroes an already-zero buffer on every replay: harmless)
        gens.append(torch.zeros(max(n * 2, 4096), dtype=torch.int32, device=dev))
    return gens[-1]


class CsrStore:
    """Per-graph CSR tables of a dataset's UNIQUE graphs, sorted once, and batches assembled from them by
    concatenation (the reference stores unique protein / drug graphs and lets pairs index
    into them, dataset/dual_dataset.py:123-125).  `collate(plan, attach_to=edge_index)` builds the batch
    tables in one small launch and memoises them on the batch's edge_index tensor, where the encoders'
    `cached_csr` finds them -- the models need no other change.

    With `edge_attr` (and `edge_types`) the store also keeps every unique graph's raw edge features ON THE DEVICE,
    rows already in dst-sorted order ("feature-table wire format"): `collate(..., table=True)` then emits an `eperm`
    that points INTO those tables, and the batch is run with `store.edge_table()` as eattr / etypes -- no per-batch
    edge-feature tensor is collated, shipped or gathered; each graph's rows are read sequentially in place."""

    def __init__(self, edge_indices, num_nodes, edge_attr=None, edge_types=None):
        """edge_indices: list of [2, E_g] int64 CUDA tensors with LOCAL node ids; num_nodes: list of ints;
        edge_attr: optional list of (e_s [E_g, 32], e_v [E_g, 1, 3]) in the graphs' original edge order;
        edge_types: optional list of int64 [E_g]."""
        csrs = [build_csr(ei, n) for ei, n in zip(edge_indices, num_nodes)]
        dev = edge_indices[0].device
        self.device = dev
        self.nodes = [int(n) for n in num_nodes]
        self.edges = [c.num_edges for c in csrs]
        self.rowptr = torch.cat([c.rowptr for c in csrs])
        self.eperm = torch.cat([c.eperm[:c.num_edges] for c in csrs] + [torch.zeros(1, dtype=torch.int32, device=dev)])
        self.esrc = torch.cat([c.esrc[:c.num_edges] for c in csrs] + [torch.zeros(1, dtype=torch.int32, device=dev)])
        self.edst = torch.cat([c.edst[:c.num_edges] for c in csrs] + [torch.zeros(1, dtype=torch.int32, device=dev)])
        self.node_off = torch.tensor([0] + list(torch.tensor(self.nodes).cumsum(0)), dtype=torch.int64, device=dev)
        self.edge_off = torch.tensor([0] + list(torch.tensor(self.edges).cumsum(0)), dtype=torch.int64, device=dev)
        self.e_s = self.e_v = self.etypes = None
        if edge_attr is not None:
            perm = [c.eperm[:c.num_edges].long() for c in csrs]
            self.e_s = torch.cat([a[0].to(dev).index_select(0, p) for a, p in zip(edge_attr, perm)]).contiguous()
            self.e_v = torch.cat([a[1].to(dev).index_select(0, p) for a, p in zip(edge_attr, perm)]).contiguous()
            if edge_types is not None:
                self.etypes = torch.cat([t.to(dev).index_select(0, p) for t, p in zip(edge_types, perm)]).contiguous()
            else:
                self.etypes = torch.zeros(self.e_s.shape[0], dtype=torch.int64, device=dev)

    def edge_table(self):
        """(eattr tuple, etypes) to hand to the encoder with batches collated with table=True."""
        if self.e_s is None:
            raise ValueError("this store was built without edge features")
        return (self.e_s, self.e_v), self.etypes

    def plan(self, ids):
        """Device-side description of a batch (graph ids in batch order): reusable across steps."""
        n = torch.tensor([self.nodes[i] for i in ids]).cumsum(0)
        e = torch.tensor([self.edges[i] for i in ids]).cumsum(0)
        dev = self.device
        return dict(sel=torch.tensor(list(ids), dtype=torch.int64, device=dev),
                    node_off=torch.cat([torch.zeros(1, dtype=torch.int64), n]).to(dev),
                    edge_off=torch.cat([torch.zeros(1, dtype=torch.int64), e]).to(dev),
                    N=int(n[-1]), E=int(e[-1]), B=len(ids))

    def collate(self, plan, attach_to=None, table=False):
        dev = self.device
        i32 = dict(dtype=torch.int32, device=dev)
        N, E = plan["N"], plan["E"]
        if table and self.e_s is None:
            raise ValueError("table=True needs a store built with edge_attr")
        rowptr = torch.empty(N + 1, **i32)
        eperm, esrc, edst = (torch.empty(max(E, 1), **i32) for _ in range(3))
        with torch.cuda.device(dev):
            rc = _lib.lib().cgvp_csr_collate(_ptr(self.rowptr), _ptr(self.eperm), _ptr(self.esrc), _ptr(self.edst),
                                             _ptr(self.node_off), _ptr(self.edge_off), _ptr(plan["sel"]),
                                             _ptr(plan["node_off"]), _ptr(plan["edge_off"]), plan["B"],
                                             1 if table else 0, _ptr(rowptr), _ptr(eperm), _ptr(esrc), _ptr(edst),
                                             _stream())
        _lib.check(rc, "cgvp_csr_collate")
        csr = Csr(rowptr, eperm, esrc, edst, N, E)
        csr.table_rows = int(self.e_s.shape[0]) if table else None
        if attach_to is not None:
            attach_to._cgvp_csr = (attach_to._version, csr)
        return csr


def shard_pairs_by_edges(edge_counts, world_size):
    """Edge-balanced assignment of a batch's pairs to ranks (the reference's own sampler balances
    batches by edge count, dataset/dual_dataset.py:476-516): longest-processing-time greedy over the per-pair protein
    edge counts.  Returns `world_size` lists of pair positions; every rank gets at least one pair when there are
    enough, and the heaviest rank carries at most (mean + the largest single pair)."""
    order = sorted(range(len(edge_counts)), key=lambda i: -int(edge_counts[i]))
    loads, parts = [0] * world_size, [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], len(parts[k])))
        parts[r].append(i)
        loads[r] += int(edge_counts[i])
    return [sorted(p) for p in parts]


CSR_CACHE_ENABLED = True
# "mfma": 16-item MFMA tiles fed from the fragment image (production path);
# "simt": one item per lane straight from the arena (cross-check / A-B timing).
VARIANT = os.environ.get("CGVP_VARIANT", "mfma")
# MFMA path: conv + node update of a layer in one launch (CGVP_FUSE_LAYER=0: two launches, for A/B timing)
FUSE_LAYER = os.environ.get("CGVP_FUSE_LAYER", "1") != "0"


def fuse_layer(num_nodes, num_edges):
    """The fused launch runs the node update on each conv wave's own targets (~30 / in-degree of them per
    16-lane tile): a win for radius graphs (3-4 edges per residue), a loss for dense kNN graphs where a
    wave owns one or two targets and would run a nearly empty node tile."""
    return FUSE_LAYER and num_edges <= 4 * num_nodes
# bench.py's roofline leg: when this is a list, every conv launch appends a
# (start, end) pair of timing events recorded on the launch stream.
KERNEL_EVENTS = None


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if KERNEL_EVENTS is not None:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()

    def __exit__(self, *exc):
        if KERNEL_EVENTS is not None:
            self.ev[1].record()
            KERNEL_EVENTS.append((self.name,) + self.ev)


def csr_memo(edge_index, num_nodes):
    """The memoised CSR of this tensor object, or None (see cached_csr)."""
    if CSR_CACHE_ENABLED:
        memo = getattr(edge_index, "_cgvp_csr", None)
        if memo is not None and memo[0] == edge_index._version and memo[1].num_nodes == num_nodes:
            return memo[1]
    return None


def cached_csr(edge_index, num_nodes, counted=None):
    """CSR of `edge_index`, memoised ON the tensor object itself (attribute
    `_cgvp_csr`), so protein layers, repeated forwards on one batch and the
    backward pass share one build.  The memo dies with the tensor and is ignored
    after any in-place write (`_version` bump); a different tensor object -- every
    freshly collated training batch -- is always rebuilt."""
    if CSR_CACHE_ENABLED:
        memo = getattr(edge_index, "_cgvp_csr", None)
        if memo is not None and memo[0] == edge_index._version and memo[1].num_nodes == num_nodes:
            return memo[1]
    csr = build_csr(edge_index, num_nodes, counted)
    if CSR_CACHE_ENABLED:
        try:
            edge_index._cgvp_csr = (edge_index._version, csr)
        except (AttributeError, RuntimeError):
            pass
    return csr


def csr_for_forward(edge_index, num_nodes, counted=None):
    """`cached_csr`, plus a forward -> backward hand-off that does not depend on CSR_CACHE_ENABLED: the tables
    of the latest forward over this tensor object are left on it for the backward of the same step (autograd
    hands the saved `edge_index` back as the same Python object in eager mode)."""
    csr = cached_csr(edge_index, num_nodes, counted)
    try:
        edge_index._cgvp_csr_step = (edge_index._version, csr)
    except (AttributeError, RuntimeError):
        pass
    return csr


def csr_for_backward(edge_index, num_nodes):
    memo = getattr(edge_index, "_cgvp_csr_step", None)
    if memo is not None and memo[0] == edge_index._version and memo[1].num_nodes == num_nodes \
            and memo[1].num_edges == int(edge_index.shape[1]) and memo[1].rowptr.device == edge_index.device:
        return memo[1]
    return cached_csr(edge_index, num_nodes)      # e.g. under torch.compile: a different tensor object


def edge_features(ca_xyz, seq_index, edge_index):
    """Protein edge features from C-alpha coordinates [N, 3] (fp32, Angstrom) and per-chain sequence indices [N]
    (utils/create_protein_features.py:225-273 on the device): -> (e_s [E, 32], e_v [E, 1, 3]) in the
    order of `edge_index` -- the `eattr` tuple the protein encoder takes."""
    ca = _f32(ca_xyz, "ca_xyz")
    seq, ei = _i64(seq_index, "seq_index"), _i64(edge_index, "edge_index")
    if ca.dim() != 2 or ca.shape[1] != 3 or seq.shape[0] != ca.shape[0] or ei.dim() != 2 or ei.shape[0] != 2:
        raise ValueError("expected ca_xyz [N, 3], seq_index [N], edge_index [2, E]")
    N, E = int(ca.shape[0]), int(ei.shape[1])
    e_s = torch.empty(E, 32, dtype=torch.float32, device=ca.device)
    e_v = torch.empty(E, 1, 3, dtype=torch.float32, device=ca.device)
    with torch.cuda.device(ca.device):
        _lib.check(_lib.lib().cgvp_edge_featurise(_ptr(ca), _ptr(seq), _ptr(ei), N, E, _ptr(e_s), _ptr(e_v), _stream()),
                   "cgvp_edge_featurise")
    return e_s, e_v


def prepare_image(params, layout, dims):
    """Fragment image of the arena for the MFMA kernels (one small launch)."""
    L = _lib.lib()
    n = int(L.cgvp_lba_image_floats(C.byref(dims), C.byref(layout)))
    if n < 0:
        _lib.check(n, "cgvp_lba_image_floats")
    image = torch.empty(n, dtype=torch.float32, device=params.device)
    with torch.cuda.device(params.device):
        _lib.check(L.cgvp_lba_prepare(C.byref(dims), C.byref(layout), _ptr(params), _ptr(image), _stream()),
                   "cgvp_lba_prepare")
    return image


def lba_encoder_forward(params, layout, dims, num_convs, x_s, x_v, ntypes, e_s, e_v, etypes, csr,
                        aggr_mean=False, return_stages=False, image=None, h0=None):
    """VectorProteinGNN_LBAModel.forward (protein_gnn.py:361-388), eval mode,
    as 1 + 2*num_convs launches: node embed, then (conv, node update) per layer
    with the output head fused into the last node update.  `h0`: the node embedding, when the caller's
    cgvp_lba_pass_begin launch already produced it (together with `image`)."""
    L = _lib.lib()
    sdt = torch.bfloat16 if dims.storage == BF16 else torch.float32
    x_s, x_v = _act(x_s, "x_s", sdt), _act(x_v, "x_v", sdt)
    e_s, e_v = _act(e_s, "eattr_s", sdt), _act(e_v, "eattr_v", sdt)
    N, E = int(x_s.shape[0]), csr.num_edges          # E = sorted edges of the batch; the feature tensors may be a resident table
    if tuple(x_s.shape) != (N, dims.node_in_s) or tuple(x_v.shape) != (N, dims.node_in_v, 3):
        raise NotImplementedError(f"node features {tuple(x_s.shape)}/{tuple(x_v.shape)} do not match the "
                                  "compiled CASTER-DTA configuration")
    rows = csr.table_rows if csr.table_rows is not None else E
    if tuple(e_s.shape) != (rows, dims.edge_in_s) or tuple(e_v.shape) != (rows, dims.edge_in_v, 3):
        raise NotImplementedError(f"edge features {tuple(e_s.shape)}/{tuple(e_v.shape)} do not match the "
                                  f"compiled CASTER-DTA configuration ({rows} rows expected)")
    if csr.num_nodes != N:
        raise ValueError("CSR tables were built for a different graph")
    if num_convs < 1:
        raise NotImplementedError("num_convs must be >= 1 (the output head is fused into the last layer)")
    nt = _i64(ntypes, "ntypes") if layout.nt_node > 0 else None
    et = _i64(etypes, "etypes") if layout.nt_edge > 0 else None
    dev = x_s.device
    h = h0 if h0 is not None else torch.empty(N, ROW, dtype=sdt, device=dev)
    h2 = torch.empty(N, ROW, dtype=sdt, device=dev)
    # edge embedding store: layer 0 writes gvp_edge + LayerNorm of every edge (sorted order), later layers read it
    e_emb = torch.empty(E + 1, EROW, dtype=sdt, device=dev) if (VARIANT == "mfma" and num_convs > 1) else None
    dh = torch.empty(N, ROW, dtype=sdt, device=dev)
    out = torch.empty(N, dims.out_s, dtype=sdt, device=dev)
    stages = {}
    if VARIANT == "mfma" and image is None:
        image = prepare_image(params, layout, dims)
    elif VARIANT != "mfma":
        image = None
    with torch.cuda.device(dev):
        st = _stream()
        d, lay, P, I = C.byref(dims), C.byref(layout), _ptr(params), _ptr(image)
        if h0 is None:
            _lib.check(L.cgvp_node_embed_fwd(d, lay, P, I, _ptr(x_s), _ptr(x_v), _ptr(nt), N, _ptr(h), None, None, st),
                       "cgvp_node_embed_fwd")
        if return_stages:
            stages["node_embed"] = h.clone()
        for layer in range(num_convs):
            if fuse_layer(N, E) and VARIANT == "mfma" and not return_stages:
                # one launch per GVPConvLayer: conv + node update (+ output head); no dh buffer in inference
                last = layer == num_convs - 1
                with _timed("conv_fwd"):
                    _lib.check(L.cgvp_conv_layer_fwd(d, lay, I, layer, _ptr(h), _ptr(e_s), _ptr(e_v), _ptr(et),
                                                     _ptr(csr.rowptr), _ptr(csr.eperm), _ptr(csr.esrc),
                                                     _ptr(csr.edst), N, E, 1 if aggr_mean else 0, C.c_void_p(0),
                                                     C.c_void_p(0), None, 1 if last else 0,
                                                     _ptr(e_emb if layer > 0 else None), _ptr(e_emb if layer == 0 else None),
                                                     C.c_void_p(0), _ptr(None if last else h2), _ptr(out), st),
                               "cgvp_conv_layer_fwd")
                if not last:
                    h, h2 = h2, h
                continue
            with _timed("conv_fwd"):
                _lib.check(L.cgvp_conv_fwd(d, lay, P, I, layer, _ptr(h), _ptr(e_s), _ptr(e_v), _ptr(et),
                                           _ptr(csr.rowptr), _ptr(csr.eperm), _ptr(csr.esrc), _ptr(csr.edst),
                                           N, E, 1 if aggr_mean else 0, _ptr(e_emb if layer > 0 else None),
                                           _ptr(e_emb if layer == 0 else None), _ptr(dh), st), "cgvp_conv_fwd")
            if return_stages:
                stages[f"conv{layer}_dh"] = dh.clone()
            last = layer == num_convs - 1
            if return_stages and last:      # materialise the last hidden state as well
                _lib.check(L.cgvp_node_update_fwd(d, lay, P, I, layer, _ptr(h), _ptr(dh), N, 0, _ptr(h2),
                                                  C.c_void_p(0), st), "cgvp_node_update_fwd")
                stages[f"conv{layer}"] = h2.clone()
            _lib.check(L.cgvp_node_update_fwd(d, lay, P, I, layer, _ptr(h), _ptr(dh), N, 1 if last else 0,
                                              _ptr(h2), _ptr(out), st), "cgvp_node_update_fwd")
            if not last:
                if return_stages:
                    stages[f"conv{layer}"] = h2.clone()
                h, h2 = h2, h
    return (out, stages) if return_stages else out


def gine_conv_forward(x, ntypes, num_ntypes, eattr, etypes, num_etypes, csr, w, cin, chid, cout, slope, mask=None,
                      rng=None):
    """One GINEConv + activation (molecule_gnn.py:260-266).  `w` maps the
    cgvp_gine_w field names to contiguous fp32 CUDA tensors."""
    L = _lib.lib()
    x = _f32(x, "x")
    eattr = _f32(eattr, "eattr")
    N = int(x.shape[0])
    nt = _i64(ntypes, "ntypes") if num_ntypes > 0 else None
    et = _i64(etypes, "etypes") if num_etypes > 0 else None
    edge_dim = int(eattr.shape[1])
    if x.shape[1] != cin - num_ntypes:
        raise ValueError(f"x has {x.shape[1]} columns, expected {cin - num_ntypes}")
    ws = {k: _f32(v, k) for k, v in w.items()}
    gw = GineW(**{k: v.data_ptr() for k, v in ws.items()})
    out = torch.empty(N, cout, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = L.cgvp_gine_conv_fwd(_ptr(x), _ptr(nt), num_ntypes, _ptr(eattr), _ptr(et), num_etypes, edge_dim,
                                  _ptr(csr.rowptr), _ptr(csr.eperm), _ptr(csr.esrc), _ptr(csr.edst), N, csr.num_edges,
                                  cin, chid, cout, C.byref(gw), float(slope), _ptr(mask), _rng_ref(rng),
                                  0 if VARIANT == "mfma" else 1, _ptr(out), _stream())
    _lib.check(rc, "cgvp_gine_conv_fwd")
    return out
