"""The encoders as `torch.library` custom ops (namespace ``caster_gvp``).

Why custom ops and not a bare ``torch.autograd.Function`` around ctypes calls: the reference wraps its model in
``torch.compile(model, dynamic=True)`` (train_model.py:422), runs the forward under ``torch.autocast``
(:561) and back-propagates a ``GradScaler``-scaled loss (:478, :570-587).  With the registrations below all three
work unchanged:

  * ``register_fake``      -- shape/dtype propagation for Dynamo / AOTAutograd (dynamic N, E);
  * ``register_autograd``  -- the backward is itself a custom op (``*_backward``), so AOTAutograd can trace it;
  * ``register_autocast``  -- float inputs are cast to fp32 (the kernels are fp32 storage / fp32 accumulate);
  * the real kernels are opaque to the compiler: each op is ONE C-ABI call (cgvp_*_forward_pass / cgvp_*_backward_pass,
    include/caster_gvp.h "WHOLE-PASS ENTRY POINTS") that issues the launch sequence of one encoder pass.

Ops
  caster_gvp::lba_encoder            VectorProteinGNN_LBAModel.forward (protein_gnn.py:361-388)
  caster_gvp::lba_encoder_backward   what autograd derives from it in the reference
  caster_gvp::gine_encoder           HomoMoleculeGNN_GINE.forward (molecule_gnn.py:254-268)
  caster_gvp::gine_encoder_backward

A training forward returns, next to the embeddings, ONE opaque workspace tensor `ws` (uint8): the fragment image of the
weights the pass ran with, every stage's inputs (node rows h_l, aggregated messages dh_l), the edge-embedding store, the
CSR tables and the pass's dropout {seed, offset} -- what the backward pass needs, laid out by cgvp_*_fwd_workspace.
The backward ops hand it back to the library together with a scratch workspace.  Dropout factors are generated inside
the kernels (Philox keyed by (seed, offset, mask id, node, channel), csrc/gvp_rng.h) from a persistent {seed, offset}
state on the device that every training pass advances inside its first kernel: no PyTorch launch per step, and a
HIP-graph replay advances it like an eager step does.

Lifetime rules (a captured HIP graph keeps writing to what it was captured with): the two workspaces are allocated per
pass from PyTorch's caching allocator -- inside a capture that is the graph's own pool; the persistent buffers -- the
CSR counters and the generator state -- are NEVER freed or replaced once handed out (`ops.csr_counters` keeps every
generation alive, `rng_state` re-seeds in place).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Tuple

import torch
from torch import Tensor

from . import _lib, ops
from .ops import _act, _f32, _i64, _ptr, _stream

MROW = 20   # dropout mask row: 16 scalar-channel + 4 vector-channel factors
# `cfg` argument of the LBA ops: the nine cgvp_dims fields, then these
_CFG_FIELDS = ("node_in_s", "node_in_v", "edge_in_s", "edge_in_v", "hidden_s", "hidden_v", "edge_hidden_s",
               "edge_hidden_v", "out_s")
CFG_NTN, CFG_NTE, CFG_NC, CFG_MEAN, CFG_LEN = 9, 10, 11, 12, 13     # the activation storage type is NOT in cfg: it follows x_s.dtype


# Test hook: a callable (count, n, width, p, device) -> [count, n, width] fp32 tensor of EXPLICIT dropout factors
# (0 or 1/(1-p)).  When set, the ops pass these masks to the kernels instead of a seed (the kernels' mask pointers
# take precedence over in-kernel generation); production leaves it None.
PINNED_MASKS = None


def draw_seed(device):
    """A fresh {seed, offset} pair: int64[2] on the device, from torch's CUDA generator."""
    return torch.randint(0, 1 << 62, (2,), dtype=torch.int64, device=device)


# Persistent generator state per (kind of encoder, device): {seed, offset} on the device.  It is drawn from torch's CUDA
# generator the first time it is needed; whenever `torch.manual_seed` / `torch.cuda.manual_seed` has changed the
# generator's seed since, it is RE-SEEDED IN PLACE (same buffer: a captured HIP graph that advances it keeps writing to
# live memory, and sees the new seed); every training pass advances `offset` ON THE DEVICE (inside the pass's first
# kernel), leaving the pass's own pair in its workspace, which its later kernels and its backward read.
_RNG_STATE = {}
_LAST_WS = {}


def rng_state(kind, device):
    key = (kind, device.index)
    gen_seed = torch.cuda.default_generators[device.index].initial_seed()
    hit = _RNG_STATE.get(key)
    if hit is None or hit[0] != gen_seed:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the dropout generator state must exist before HIP-graph capture: run one training "
                               "step eagerly first (any warm-up does)")
        if hit is None:
            hit = [gen_seed, draw_seed(device)]
            _RNG_STATE[key] = hit
        else:
            hit[1].copy_(draw_seed(device))
            hit[0] = gen_seed
    return hit[1]


def last_seed(kind):
    """The {seed, offset} pair (int64[2], a view into the pass's workspace) of the latest training pass of `kind`
    ('lba' / 'gine') -- tests export the factors the kernels used from it (ops.dropout_masks)."""
    ws, off = _LAST_WS[kind]
    return ws[off:off + 16].view(torch.int64)


def make_cfg(dims, num_ntypes, num_etypes, num_convs, aggr_mean):
    return [int(getattr(dims, f)) for f in _CFG_FIELDS] + [int(num_ntypes), int(num_etypes), int(num_convs),
                                                           1 if aggr_mean else 0]


_DIMS = {}


def _dims_layout(cfg, sdt=torch.float32):
    """(cgvp_dims, cgvp_layout, fragment-image floats) of a configuration, built once."""
    key = (tuple(cfg[:CFG_MEAN]), sdt)
    hit = _DIMS.get(key)
    if hit is None:
        dims = ops.make_dims(storage=ops.BF16 if sdt == torch.bfloat16 else ops.F32,
                             **{f: cfg[i] for i, f in enumerate(_CFG_FIELDS)})
        layout = ops.lba_layout(dims, cfg[CFG_NTN], cfg[CFG_NTE], cfg[CFG_NC])
        img = int(_lib.lib().cgvp_lba_image_floats(C.byref(dims), C.byref(layout)))
        if img < 0:
            _lib.check(img, "cgvp_lba_image_floats")
        hit = _DIMS[key] = (dims, layout, img)
    return hit


def flat_arena(params):
    """The arena-ordered parameter list as ONE flat fp32 buffer: zero-copy when the tensors are consecutive
    views of one storage (what gvp_hip.arena.ParamArena maintains), one `cat` otherwise."""
    p0 = params[0]
    st, off = p0.untyped_storage(), p0.storage_offset()
    o = off
    for p in params:
        if p.untyped_storage().data_ptr() != st.data_ptr() or p.storage_offset() != o or not p.is_contiguous():
            return torch.cat([q.detach().reshape(-1) for q in params])
        o += p.numel()
    return torch.empty(0, dtype=p0.dtype, device=p0.device).set_(st, off, (o - off,), (1,))


# ------------------------------------------------------------------------------------------------ workspace sizes
# Python mirrors of cgvp_lba_fwd_workspace / cgvp_gine_fwd_workspace (csrc/pass_api.hip) for the fake kernels, which
# must produce the real shapes from symbolic N, E without calling into the library (tests/test_pass_api.py compares
# them with the C functions).
def _up(v, a=256):
    return (v + a - 1) // a * a


def lba_fwd_ws_bytes(N, E, num_convs, image_floats, esize, save_state):
    rows = 2 * num_convs + 1 if save_state else 3
    e1 = E if E > 0 else 1
    return (_up(16) + _up(image_floats * 4) + _up(rows * (N + N % 2) * ops.ROW * esize) + _up((E + 1) * ops.EROW * esize)
            + _up((N + 1) * 4) + 3 * _up(e1 * 4))


def gine_fwd_ws_bytes(N, E, widths, save_state, saved_only=False):
    """total bytes of a GINE forward workspace; saved_only: the prefix the backward reads (everything but the CSR build's
    scratch of edge ids in arrival order, which is not a function of the inputs)."""
    L = len(widths) - 1
    e1 = E if E > 0 else 1
    if save_state:
        hidden = sum(_up(N * widths[l + 1] * 4) for l in range(L - 1))
        # + per layer: the aggregated messages [N][ceil16(width)] fp32 and 8 B of ReLU pattern per sorted edge (ABI v29)
        hidden += sum(_up(N * ((widths[l] + 15) // 16 * 16) * 4) + _up(e1 * 8) for l in range(L))
    else:
        hidden = 2 * _up(N * max([1] + list(widths[1:L])) * 4)
    return _up(16) + hidden + _up((N + 1) * 4) + (3 if saved_only else 4) * _up(e1 * 4)


def _csr_ptrs(csr):
    """(rowptr, eperm, esrc, edst) data pointers of an op's `csr` argument (4 int32 tensors, or empty)."""
    if len(csr) == 0:
        return 0, 0, 0, 0
    if len(csr) != 4 or any(t.dtype != torch.int32 or not t.is_cuda for t in csr):
        raise ValueError("csr must be [] or the four int32 CUDA tables rowptr, eperm, esrc, edst")
    return tuple(t.data_ptr() for t in csr)


def _lba_batch(x_s, x_v, nt, e_s, e_v, et, edge_index, csr, N, E):
    rp, ep, es_, ed = _csr_ptrs(csr)
    return _lib.LbaBatch(N, E, x_s.data_ptr(), x_v.data_ptr(), nt.data_ptr() if nt is not None else 0,
                         e_s.data_ptr() if E > 0 else 0, e_v.data_ptr() if E > 0 else 0,
                         et.data_ptr() if (et is not None and E > 0) else 0,
                         edge_index.data_ptr() if (E > 0 and not csr) else 0, rp, ep, es_, ed)


# ===================================================================================== protein encoder
@torch.library.custom_op("caster_gvp::lba_encoder", mutates_args=(), device_types="cuda")
def lba_encoder_op(params: List[Tensor], x_s: Tensor, x_v: Tensor, ntypes: Tensor, e_s: Tensor, e_v: Tensor,
                   etypes: Tensor, edge_index: Tensor, csr: List[Tensor], cfg: List[int], dropout_p: float,
                   save_state: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (out [N, out_s], ws uint8[...]: the pass's forward workspace (cgvp_lba_fwd_ws layout; empty when save_state is
    False), masks (empty unless the PINNED_MASKS test hook is set: [2 L, N, 20])).
    `csr`: [] (the pass builds the dst-sorted tables from edge_index) or [rowptr, eperm, esrc, edst] (tables collated
    from a CsrStore / memoised on the batch).  save_state=False is the inference launch sequence."""
    if ops.VARIANT != "mfma":
        if save_state:
            raise NotImplementedError("training / gradients need the MFMA kernels (CGVP_VARIANT=mfma)")
        return _lba_simt_inference(params, x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, csr, cfg)
    out, ws, masks = lba_forward(flat_arena(params), x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, csr, cfg, dropout_p,
                                 save_state, _WS_ALLOC)
    if not save_state:
        ws = ws.new_empty(0)
    return out, ws, masks


def _lba_simt_inference(params, x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, csr, cfg):
    """CGVP_VARIANT=simt: the independent one-item-per-lane kernels, through the fine-grained entry points."""
    sdt = ops.storage_dtype(x_s)
    dims, layout, _ = _dims_layout(cfg, sdt)
    N = int(x_s.shape[0])
    tables = ops.Csr(*csr, N, int(edge_index.shape[1])) if csr else ops.build_csr(edge_index, N)
    out = ops.lba_encoder_forward(flat_arena(params), layout, dims, cfg[CFG_NC], x_s, x_v, ntypes, e_s, e_v, etypes, tables,
                                  aggr_mean=bool(cfg[CFG_MEAN]))
    e8 = torch.empty(0, dtype=torch.uint8, device=x_s.device)
    return out, e8, torch.empty(0, dtype=torch.float32, device=x_s.device)


def lba_forward(flat, x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, csr, cfg, dropout_p, save_state, alloc=torch.empty):
    """One cgvp_lba_forward_pass.  `alloc`: torch.empty (the workspace's alignment gaps stay uninitialised) or torch.zeros
    (the custom op: its outputs must be a pure function of its inputs for opcheck / compiled-vs-eager comparisons)."""
    L = _lib.lib()
    sdt = ops.storage_dtype(x_s)                       # activation storage: bf16 when the features arrive in bf16
    dims, layout, img = _dims_layout(cfg, sdt)
    nc, mean = cfg[CFG_NC], bool(cfg[CFG_MEAN])
    x_s, x_v, e_s, e_v = _act(x_s, "x_s", sdt), _act(x_v, "x_v", sdt), _act(e_s, "eattr_s", sdt), _act(e_v, "eattr_v", sdt)
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError(f"edge_index must be [2, E], got {tuple(edge_index.shape)}")
    N, E = int(x_s.shape[0]), int(edge_index.shape[1])      # e_s / e_v may be a resident feature table (CsrStore table mode)
    dev = x_s.device
    if flat.numel() != layout.total:
        raise RuntimeError(f"parameter arena has {flat.numel()} floats, kernels expect {layout.total}")
    if tuple(x_s.shape) != (N, dims.node_in_s) or tuple(x_v.shape) != (N, dims.node_in_v, 3) \
            or tuple(e_s.shape[1:]) != (dims.edge_in_s,) or tuple(e_v.shape[1:]) != (dims.edge_in_v, 3) \
            or e_s.shape[0] != e_v.shape[0] or (not csr and e_s.shape[0] != E):
        raise NotImplementedError("feature shapes do not match the compiled CASTER-DTA configuration")
    nt = _i64(ntypes, "ntypes") if layout.nt_node > 0 else None
    et = _i64(etypes, "etypes") if layout.nt_edge > 0 else None
    ei = _i64(edge_index, "edge_index") if not csr else edge_index
    masks = torch.empty(0, dtype=torch.float32, device=dev)
    draw = False
    if dropout_p > 0 and save_state:
        if PINNED_MASKS is not None:
            masks = PINNED_MASKS(2 * nc, N, MROW, dropout_p, dev)
        else:
            draw = True
    es = 2 if sdt == torch.bfloat16 else 4
    ws = alloc(lba_fwd_ws_bytes(N, E, nc, img, es, save_state), dtype=torch.uint8, device=dev)
    out = torch.empty(N, dims.out_s, dtype=sdt, device=dev)
    batch = _lba_batch(x_s, x_v, nt, e_s, e_v, et, ei, csr, N, E)
    with torch.cuda.device(dev):
        counters = ops.csr_counters(dev, N) if not csr else None
        rc = L.cgvp_lba_forward_pass(C.byref(dims), C.byref(layout), _ptr(flat), C.byref(batch), 1 if mean else 0,
                                     float(dropout_p) if save_state else 0.0,
                                     _ptr(rng_state("lba", dev)) if draw else None, _ptr(masks), _ptr(counters), _ptr(ws),
                                     1 if save_state else 0, 0 if ops.FUSE_LAYER else 1, _ptr(out), _stream())
    if rc != 0 and counters is not None:
        counters.zero_()               # the persistent counters must not stay half-used (the next build assumes zeros)
    _lib.check(rc, "cgvp_lba_forward_pass")
    if draw:
        _LAST_WS["lba"] = (ws, 0)
    return out, ws, masks


@lba_encoder_op.register_fake
def _(params, x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, csr, cfg, dropout_p, save_state):
    N, E, nc = x_s.shape[0], edge_index.shape[1], cfg[CFG_NC]
    sdt = torch.bfloat16 if x_s.dtype == torch.bfloat16 else torch.float32
    out = x_s.new_empty((N, cfg[8]), dtype=sdt)
    e32 = x_s.new_empty((0,), dtype=torch.float32)
    if not save_state:
        return out, x_s.new_empty((0,), dtype=torch.uint8), e32
    pinned = dropout_p > 0 and PINNED_MASKS is not None
    masks = x_s.new_empty((2 * nc, N, MROW), dtype=torch.float32) if pinned else e32
    img = _image_floats_for(cfg, sdt)
    ws = x_s.new_empty((lba_fwd_ws_bytes(N, E, nc, img, 2 if sdt == torch.bfloat16 else 4, True),), dtype=torch.uint8)
    return out, ws, masks


# The custom ops return the forward workspace (saved for the backward op), and an op's outputs must be a pure function
# of its inputs for opcheck / compiled-vs-eager comparisons: alignment gaps, the odd-N padding row of the state blocks
# and the rows of dropped edges are written by no kernel, so the workspace is zero-filled first.  That fill (20 MB at
# davis_b64) costs 10 us of a 260 us captured step on this host path (measured: 0.2600 vs 0.2493 ms; the C++ eager
# nodes allocate with empty).  CGVP_WS_EMPTY=1 trades the bitwise reproducibility of those unread bytes for the 10 us.
_WS_ALLOC = torch.empty if os.environ.get("CGVP_WS_EMPTY") == "1" else torch.zeros


def _image_floats_for(cfg, sdt):
    """Fragment-image length of a configuration: a host-only library call (no GPU), cached."""
    return _dims_layout([int(c) for c in cfg], sdt)[2]


@torch.library.custom_op("caster_gvp::lba_encoder_backward", mutates_args=(), device_types="cuda")
def lba_encoder_backward_op(g_out: Tensor, x_s: Tensor, x_v: Tensor, ntypes: Tensor, e_s: Tensor, e_v: Tensor,
                            etypes: Tensor, edge_index: Tensor, csr: List[Tensor], ws: Tensor, masks: Tensor,
                            cfg: List[int], dropout_p: float, need_x: bool,
                            need_e: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> (grad arena [layout.total], g_x_s [N, 17], g_x_v [N, 3, 3], g_e_s [E, 32], g_e_v [E, 1, 3]) -- the feature
    gradients are empty unless need_x / need_e."""
    return lba_backward(g_out, x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, csr, ws, masks, cfg, dropout_p, need_x, need_e)


def lba_backward(g_out, x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, csr, ws, masks, cfg, dropout_p, need_x,
                 need_e=False):
    """One cgvp_lba_backward_pass."""
    L = _lib.lib()
    sdt = ops.storage_dtype(x_s)
    dims, layout, _ = _dims_layout(cfg, sdt)
    mean = bool(cfg[CFG_MEAN])
    x_s, x_v, e_s, e_v = _act(x_s, "x_s", sdt), _act(x_v, "x_v", sdt), _act(e_s, "eattr_s", sdt), _act(e_v, "eattr_v", sdt)
    N, E = int(x_s.shape[0]), int(edge_index.shape[1])
    dev = x_s.device
    nt = _i64(ntypes, "ntypes") if layout.nt_node > 0 else None
    et = _i64(etypes, "etypes") if layout.nt_edge > 0 else None
    g_out = _f32(g_out.float(), "grad_output")               # every gradient buffer is fp32, whatever the storage type
    f32 = dict(dtype=torch.float32, device=dev)
    gparams = torch.empty(layout.total, **f32)               # every element is STORED by the pass's final reduce
    nbytes = int(L.cgvp_lba_bwd_workspace_bytes(C.byref(dims), C.byref(layout), N, E))
    if nbytes < 0:
        _lib.check(nbytes, "cgvp_lba_bwd_workspace_bytes")
    bws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    g_x_s = torch.empty(N, dims.node_in_s, **f32) if need_x else None
    g_x_v = torch.empty(N, dims.node_in_v, 3, **f32) if need_x else None
    if need_e and int(e_s.shape[0]) != E:
        raise NotImplementedError("gradients w.r.t. the edge features of a resident feature table are not produced")
    # rows of valid edges are STORED by the edge stage, the rest zero-filled by the pass (no edges: no edge stage)
    g_e_s = (torch.empty if E > 0 else torch.zeros)(E, dims.edge_in_s, **f32) if need_e else None
    g_e_v = (torch.empty if E > 0 else torch.zeros)(E, dims.edge_in_v, 3, **f32) if need_e else None
    batch = _lba_batch(x_s, x_v, nt, e_s, e_v, et, edge_index, csr, N, E)
    if ws.data_ptr() % 256:
        ws = ws.clone()
    with torch.cuda.device(dev):
        rc = L.cgvp_lba_backward_pass(C.byref(dims), C.byref(layout), C.byref(batch), 1 if mean else 0, float(dropout_p),
                                      _ptr(masks), _ptr(ws), _ptr(g_out), _ptr(bws), _ptr(gparams), _ptr(g_x_s),
                                      _ptr(g_x_v), _ptr(g_e_s if E > 0 else None), _ptr(g_e_v if E > 0 else None), _stream())
    _lib.check(rc, "cgvp_lba_backward_pass")
    none = lambda: torch.empty(0, **f32)                     # (outputs of a custom op must not alias each other)
    return (gparams, g_x_s if need_x else none(), g_x_v if need_x else none(), g_e_s if need_e else none(),
            g_e_v if need_e else none())


@lba_encoder_backward_op.register_fake
def _(g_out, x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, csr, ws, masks, cfg, dropout_p, need_x, need_e):
    sdt = torch.bfloat16 if x_s.dtype == torch.bfloat16 else torch.float32
    total = _dims_layout([int(c) for c in cfg], sdt)[1].total
    gp = x_s.new_empty((total,), dtype=torch.float32)
    none = lambda: x_s.new_empty((0,), dtype=torch.float32)
    f32 = dict(dtype=torch.float32)
    gx = (x_s.new_empty(x_s.shape, **f32), x_v.new_empty(x_v.shape, **f32)) if need_x else (none(), none())
    ge = (e_s.new_empty(e_s.shape, **f32), e_v.new_empty(e_v.shape, **f32)) if need_e else (none(), none())
    return (gp, *gx, *ge)


def _lba_setup(ctx, inputs, output):
    params, x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, csr, cfg, dropout_p, save_state = inputs
    out, ws, masks = output
    if not save_state:
        raise RuntimeError("caster_gvp::lba_encoder was run with save_state=False; gradients need save_state=True")
    ctx.cfg, ctx.dropout_p, ctx.ncsr = cfg, dropout_p, len(csr)
    ctx.shapes = [tuple(p.shape) for p in params]
    ctx.set_materialize_grads(False)      # ws is an output only to be saved: no zero-filled "gradient" for it
    ctx.save_for_backward(x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, ws, masks, *csr)


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


def _lba_backward(ctx, g_out, g_ws, g_masks):
    if g_out is None:
        return (None,) * 8 + ([None] * ctx.ncsr, None, None, None)
    x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, ws, masks, *csr = ctx.saved_tensors
    need_x = bool(ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
    need_e = bool(ctx.needs_input_grad[4] or ctx.needs_input_grad[5])
    gflat, g_x_s, g_x_v, g_e_s, g_e_v = torch.ops.caster_gvp.lba_encoder_backward(
        g_out.contiguous(), x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, list(csr), ws, masks, ctx.cfg,
        ctx.dropout_p, need_x, need_e)
    grads = [g.view(s) for g, s in zip(torch.split(gflat, [_numel(s) for s in ctx.shapes]), ctx.shapes)]
    if need_x and x_s.dtype != torch.float32:                 # input gradients are produced in fp32
        g_x_s, g_x_v = g_x_s.to(x_s.dtype), g_x_v.to(x_v.dtype)
    if need_e and e_s.dtype != torch.float32:
        g_e_s, g_e_v = g_e_s.to(e_s.dtype), g_e_v.to(e_v.dtype)
    return (grads, g_x_s if need_x else None, g_x_v if need_x else None, None, g_e_s if need_e else None,
            g_e_v if need_e else None, None, None, [None] * ctx.ncsr, None, None, None)


torch.library.register_autograd("caster_gvp::lba_encoder", _lba_backward, setup_context=_lba_setup)
torch.library.register_autocast("caster_gvp::lba_encoder", "cuda", torch.float32)


def _memo_tables(edge_index, num_nodes):
    """[rowptr, eperm, esrc, edst] of a CSR memoised on this tensor object (CsrStore.collate(attach_to=...),
    ops.cached_csr), or [] -- the pass then builds the tables itself."""
    if torch.compiler.is_compiling():
        return []
    memo = ops.csr_memo(edge_index, num_nodes)
    if memo is None or memo.num_edges != int(edge_index.shape[1]):
        return []
    return [memo.rowptr, memo.eperm, memo.esrc, memo.edst]


def lba_encoder(model, x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, train_dropout, save_state):
    """VectorProteinGNN_LBAModel.forward through the custom op (autograd, dropout when training)."""
    # plain Python ints only (no ctypes objects here: this function is traced by Dynamo under torch.compile);
    # the op validates them against the compiled kernel configuration
    cfg = [model.in_channels[0], model.in_channels[1], model.edge_dim[0], model.edge_dim[1],
           model.hidden_channels[0], model.hidden_channels[1], model.edge_hidden_channels[0],
           model.edge_hidden_channels[1], model.out_channels[0], model.num_ntypes, model.num_etypes,
           model.num_convs, 1 if model.aggr == "mean" else 0]
    p = float(model.dropout_rate) if train_dropout else 0.0
    br = _eager_bridge()
    if br is not None and torch.is_grad_enabled() and (e_s.requires_grad or e_v.requires_grad):
        br = None       # gradients w.r.t. the raw edge features (rare: attribution studies): the custom-op path returns them
    if br is not None and torch.is_autocast_enabled("cuda"):
        # train_model.py:561 runs the forward under autocast: what the custom ops' autocast rule does (float inputs in fp32;
        # the kernels are fp32 storage / fp32 accumulate there), done here so that the eager loop keeps the C++ fast path
        x_s, x_v, e_s, e_v = (t.float() if t.dtype in (torch.float16, torch.bfloat16) else t for t in (x_s, x_v, e_s, e_v))
    if br is not None:
        # eager mode: the C++ autograd node (csrc/torch_bridge.cpp) -- same C entry points, a fraction of the host time
        dev, N = x_s.device, x_s.shape[0]
        csr = _memo_tables(edge_index, N)
        masks, draw = None, False
        if p > 0 and save_state:
            if PINNED_MASKS is not None:
                masks = PINNED_MASKS(2 * model.num_convs, N, MROW, p, dev)
            else:
                draw = True
        params = model.op_params()
        with torch.cuda.device(dev):
            out, ws, zero_copy = br.lba_encoder(params, x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, csr, cfg, p,
                                                save_state, masks, rng_state("lba", dev) if draw else None,
                                                None if csr else ops.csr_counters(dev, N), ops.FUSE_LAYER)
        if draw:
            _LAST_WS["lba"] = (ws, 0)
        if not zero_copy and not model._fused and model._onehot_ntypes and model._onehot_etypes:
            model._arena.rebuild()          # something re-materialised the parameters: re-seat them as arena views
        return out
    # The custom-op path (torch.compile; CGVP_BRIDGE=0): the op takes the parameter ARENA as one tensor -- a 50-tensor
    # list costs ~0.3 ms of torch.library marshalling per call (forward, and again for the split gradients), measured on
    # the CPU.  `cat` keeps autograd's connection to the leaves: under torch.compile Inductor emits one small copy kernel
    # and the backward's split is 50 views; in eager mode it is one extra launch.
    params = model.op_params()
    arena = params[0] if len(params) == 1 else torch.cat([q.reshape(-1) for q in params])
    out, _, _ = torch.ops.caster_gvp.lba_encoder(
        [arena], x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, _memo_tables(edge_index, x_s.shape[0]), cfg,
        p, save_state)
    return out


def _eager_bridge():
    """The C++ fast path, when it applies: eager mode (Dynamo traces the custom ops instead), MFMA kernels.  (Under autocast
    the callers cast half-precision inputs to fp32 first, which is what the custom ops' autocast rule does.)"""
    if torch.compiler.is_compiling() or ops.VARIANT != "mfma":
        return None
    return _lib.bridge()


# ===================================================================================== drug encoder
_GINE_KEYS = ("eps", "w0", "b0", "w1", "b1", "we", "be")     # slab / state_dict order of one GINEConv


def _gine_cfg(widths, num_ntypes, num_etypes, edge_dim, slope):
    nl = len(widths) - 1
    if nl < 1 or nl > _lib.GINE_MAX_LAYERS:
        raise NotImplementedError(f"the GINE pass supports 1..{_lib.GINE_MAX_LAYERS} layers")
    w = (C.c_int32 * (_lib.GINE_MAX_LAYERS + 1))(*([int(v) for v in widths] + [0] * (_lib.GINE_MAX_LAYERS - nl)))
    return _lib.GineCfg(nl, w, int(num_ntypes), int(num_etypes), int(edge_dim), float(slope))


def _gine_split(flat, widths, num_ntypes, num_etypes, edge_dim):
    """The 7 L kernel tensors as views of ONE flat tensor laid out in _GINE_KEYS order per layer (what the dispatcher
    passes under torch.compile: a one-element parameter list costs a fraction of a 14-tensor list in torch.library)."""
    ke = num_etypes + edge_dim
    out, off = [], 0
    for l in range(len(widths) - 1):
        cin, ch = widths[l], widths[l + 1]
        for shape in ((1,), (ch, cin), (ch,), (ch, ch), (ch,), (cin, ke), (cin,)):
            n = 1
            for v in shape:
                n *= v
            out.append(flat[off:off + n].view(shape))
            off += n
    if off != flat.numel():
        raise RuntimeError(f"flat GINE parameter tensor has {flat.numel()} floats, the layers need {off}")
    return out


def _gine_weights(params, nl):
    keep = [_f32(p, "weight") for p in params]
    arr = (_lib.GineW * nl)()
    for l in range(nl):
        for k, t in zip(_GINE_KEYS, keep[7 * l:7 * l + 7]):
            setattr(arr[l], k, t.data_ptr())
    return arr, keep


def _gine_batch(x, nt, eattr, et, edge_index, csr, N, E):
    rp, ep, es_, ed = _csr_ptrs(csr)
    return _lib.GineBatch(N, E, x.data_ptr(), nt.data_ptr() if nt is not None else 0, eattr.data_ptr() if E > 0 else 0,
                          et.data_ptr() if (et is not None and E > 0) else 0,
                          edge_index.data_ptr() if (E > 0 and not csr) else 0, rp, ep, es_, ed)


def _mask_ptrs(masks, nl):
    live = [m for m in masks if m.numel()]
    if not live:
        return None
    return (C.c_void_p * max(nl - 1, 1))(*[m.data_ptr() if m.numel() else 0 for m in masks])


@torch.library.custom_op("caster_gvp::gine_encoder", mutates_args=(), device_types="cuda")
def gine_encoder_op(params: List[Tensor], x: Tensor, ntypes: Tensor, eattr: Tensor, etypes: Tensor,
                    edge_index: Tensor, csr: List[Tensor], widths: List[int], num_ntypes: int, num_etypes: int,
                    slope: float, dropout_p: float, save_state: bool) -> Tuple[Tensor, Tensor, List[Tensor]]:
    """-> (out [N, widths[-1]], ws uint8[...]: forward workspace (cgvp_gine_fwd_ws layout: hidden activations, CSR
    tables, dropout seed; empty when save_state is False), masks [m_0 .. m_{L-2}] (PINNED_MASKS test hook only; empty
    tensors otherwise))."""
    out, ws, masks = gine_forward(params, x, ntypes, eattr, etypes, edge_index, csr, widths, num_ntypes, num_etypes, slope,
                                  dropout_p, save_state, _WS_ALLOC)
    if not save_state:
        return out, ws.new_empty(0), []
    # what the op hands on is the part the backward reads: the scratch behind it holds edge ids in arrival order
    return out, ws[:gine_fwd_ws_bytes(int(x.shape[0]), int(edge_index.shape[1]), widths, True, saved_only=True)], masks


def gine_forward(params, x, ntypes, eattr, etypes, edge_index, csr, widths, num_ntypes, num_etypes, slope, dropout_p,
                 save_state, alloc=torch.empty):
    """One cgvp_gine_forward_pass."""
    L = _lib.lib()
    nl = len(widths) - 1
    x, eattr = _f32(x, "x"), _f32(eattr, "eattr")
    N, E, dev = int(x.shape[0]), int(edge_index.shape[1]), x.device
    edge_dim = int(eattr.shape[1])
    if x.shape[1] != widths[0] - num_ntypes:
        raise ValueError(f"x has {x.shape[1]} columns, expected {widths[0] - num_ntypes}")
    cfg = _gine_cfg(widths, num_ntypes, num_etypes, edge_dim, slope)
    if len(params) == 1 and 7 * nl > 1:
        params = _gine_split(_f32(params[0], "weights"), widths, num_ntypes, num_etypes, edge_dim)
    warr, keep = _gine_weights(params, nl)
    nt = _i64(ntypes, "ntypes") if num_ntypes > 0 else None
    et = _i64(etypes, "etypes") if num_etypes > 0 else None
    ei = _i64(edge_index, "edge_index") if not csr else edge_index
    masks, draw = [], False
    if dropout_p > 0 and nl > 1 and save_state:
        if PINNED_MASKS is not None:
            masks = [PINNED_MASKS(1, N, widths[l + 1], dropout_p, dev)[0].contiguous() for l in range(nl - 1)]
        else:
            draw = True
    if not masks:
        masks = [x.new_empty(0) for _ in range(nl - 1)]
    ws = alloc(gine_fwd_ws_bytes(N, E, widths, save_state), dtype=torch.uint8, device=dev)
    out = torch.empty(N, widths[-1], dtype=torch.float32, device=dev)
    batch = _gine_batch(x, nt, eattr, et, ei, csr, N, E)
    with torch.cuda.device(dev):
        counters = ops.csr_counters(dev, N) if not csr else None
        rc = L.cgvp_gine_forward_pass(C.byref(cfg), warr, C.byref(batch), float(dropout_p) if save_state else 0.0,
                                      _ptr(rng_state("gine", dev)) if draw else None, _mask_ptrs(masks, nl), _ptr(counters),
                                      _ptr(ws), 1 if save_state else 0, 0 if ops.VARIANT == "mfma" else 1, _ptr(out),
                                      _stream())
    if rc != 0 and counters is not None:
        counters.zero_()
    _lib.check(rc, "cgvp_gine_forward_pass")
    if draw:
        _LAST_WS["gine"] = (ws, 0)
    return out, ws, masks


@gine_encoder_op.register_fake
def _(params, x, ntypes, eattr, etypes, edge_index, csr, widths, num_ntypes, num_etypes, slope, dropout_p, save_state):
    N, E, nl = x.shape[0], edge_index.shape[1], len(widths) - 1
    out = x.new_empty((N, widths[-1]), dtype=torch.float32)
    if not save_state:
        return out, x.new_empty((0,), dtype=torch.uint8), []
    pinned = dropout_p > 0 and nl > 1 and PINNED_MASKS is not None
    masks = [x.new_empty((N, widths[l + 1]) if pinned else (0,), dtype=torch.float32) for l in range(nl - 1)]
    ws = x.new_empty((gine_fwd_ws_bytes(N, E, widths, True, saved_only=True),), dtype=torch.uint8)
    return out, ws, masks


@torch.library.custom_op("caster_gvp::gine_encoder_backward", mutates_args=(), device_types="cuda")
def gine_encoder_backward_op(g_out: Tensor, params: List[Tensor], x: Tensor, ntypes: Tensor, eattr: Tensor,
                             etypes: Tensor, edge_index: Tensor, csr: List[Tensor], ws: Tensor, masks: List[Tensor],
                             widths: List[int], num_ntypes: int, num_etypes: int, slope: float,
                             dropout_p: float, need_x: bool, bwd_workgroups: int) -> Tuple[Tensor, Tensor]:
    """-> (the gradients of `params`, flattened and concatenated in that order, g_x (empty unless need_x))."""
    return gine_backward(g_out, params, x, ntypes, eattr, etypes, edge_index, csr, ws, masks, widths, num_ntypes,
                         num_etypes, slope, dropout_p, need_x, bwd_workgroups)


def gine_backward(g_out, params, x, ntypes, eattr, etypes, edge_index, csr, ws, masks, widths, num_ntypes, num_etypes,
                  slope, dropout_p, need_x, bwd_workgroups):
    """One cgvp_gine_backward_pass."""
    L = _lib.lib()
    nl = len(widths) - 1
    x, eattr = _f32(x, "x"), _f32(eattr, "eattr")
    N, E, dev = int(x.shape[0]), int(edge_index.shape[1]), x.device
    f32 = dict(dtype=torch.float32, device=dev)
    cfg = _gine_cfg(widths, num_ntypes, num_etypes, int(eattr.shape[1]), slope)
    if len(params) == 1 and 7 * nl > 1:
        params = _gine_split(_f32(params[0], "weights"), widths, num_ntypes, num_etypes, int(eattr.shape[1]))
    warr, keep = _gine_weights(params, nl)
    nt = _i64(ntypes, "ntypes") if num_ntypes > 0 else None
    et = _i64(etypes, "etypes") if num_etypes > 0 else None
    g = _f32(g_out, "grad_output")
    gflat = torch.empty(sum(p.numel() for p in params), **f32)        # every element is STORED by the pass's reduce
    nbytes = int(L.cgvp_gine_bwd_workspace_bytes(C.byref(cfg), N, E))
    if nbytes < 0:
        _lib.check(nbytes, "cgvp_gine_bwd_workspace_bytes")
    bws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    g_x = torch.empty(N, widths[0] - num_ntypes, **f32) if need_x else None
    batch = _gine_batch(x, nt, eattr, et, edge_index, csr, N, E)
    if ws.data_ptr() % 256:
        ws = ws.clone()
    with torch.cuda.device(dev):
        rc = L.cgvp_gine_backward_pass(C.byref(cfg), warr, C.byref(batch), float(dropout_p), _mask_ptrs(masks, nl), _ptr(ws),
                                       _ptr(g), _ptr(bws), _ptr(gflat), _ptr(g_x), int(bwd_workgroups), _stream())
    _lib.check(rc, "cgvp_gine_backward_pass")
    return gflat, (g_x if need_x else torch.empty(0, **f32))


@gine_encoder_backward_op.register_fake
def _(g_out, params, x, ntypes, eattr, etypes, edge_index, csr, ws, masks, widths, num_ntypes, num_etypes, slope,
      dropout_p, need_x, bwd_workgroups):
    gflat = x.new_empty((sum(p.numel() for p in params),), dtype=torch.float32)
    return gflat, (x.new_empty(x.shape, dtype=torch.float32) if need_x else x.new_empty((0,), dtype=torch.float32))


def _gine_setup(ctx, inputs, output):
    (params, x, ntypes, eattr, etypes, edge_index, csr, widths, num_ntypes, num_etypes, slope, dropout_p,
     save_state) = inputs
    out, ws, masks = output
    if not save_state:
        raise RuntimeError("caster_gvp::gine_encoder was run with save_state=False; gradients need save_state=True")
    ctx.meta = (widths, num_ntypes, num_etypes, slope, dropout_p, len(params), len(csr))
    ctx.shapes = [tuple(p.shape) for p in params]
    ctx.set_materialize_grads(False)
    ctx.save_for_backward(x, ntypes, eattr, etypes, edge_index, ws, *params, *csr, *masks)


def _gine_backward(ctx, g_out, g_ws, g_masks):
    widths, num_ntypes, num_etypes, slope, dropout_p, n_params, n_csr = ctx.meta
    if g_out is None:
        return (None,) * 6 + ([None] * n_csr,) + (None,) * 6
    x, ntypes, eattr, etypes, edge_index, ws, *rest = ctx.saved_tensors
    params, csr, masks = rest[:n_params], rest[n_params:n_params + n_csr], rest[n_params + n_csr:]
    need_x = bool(ctx.needs_input_grad[1])
    gflat, g_x = torch.ops.caster_gvp.gine_encoder_backward(
        g_out.contiguous(), params, x, ntypes, eattr, etypes, edge_index, list(csr), ws, list(masks), widths, num_ntypes,
        num_etypes, slope, dropout_p, need_x, GINE_BWD_WORKGROUPS)
    grads = [g.view(s) for g, s in zip(torch.split(gflat, [_numel(s) for s in ctx.shapes]), ctx.shapes)]
    return (grads, g_x if need_x else None, None, None, None, None, [None] * n_csr, None, None, None, None, None, None)


# Cap on the CUs of the GINE backward (0 = the library default of 16: inside JointGNN it runs beside the
# protein backward, whose kernels own 240 of the 256 CUs).  A host that trains the drug encoder alone may
# raise it; it is forwarded as the `max_workgroups` ARGUMENT of cgvp_gine_backward_pass (the library has no state).
GINE_BWD_WORKGROUPS = 0

torch.library.register_autograd("caster_gvp::gine_encoder", _gine_backward, setup_context=_gine_setup)
torch.library.register_autocast("caster_gvp::gine_encoder", "cuda", torch.float32)


def gine_params(model, one_leaf_ok=False):
    """The 7 L tensors the kernels take, in _GINE_KEYS order.  A model in one-leaf mode (`fuse_parameters`): the arena
    itself for the C++ fast path (`one_leaf_ok`), else autograd-connected views of it."""
    if getattr(model, "_fused", False):
        if one_leaf_ok:
            return [model.arena]
        return [v for _, v in model._leaf.views(model.arena)]
    emb = None if model._onehot_etypes else model.etype_embedding
    params = []
    for conv in model.conv_list:
        kw = conv.kernel_weights(emb)
        params += [kw[k] for k in _GINE_KEYS]
    return params


def gine_encoder(model, x, ntypes, eattr, etypes, edge_index, slope, train_dropout, save_state):
    """HomoMoleculeGNN_GINE.forward through the custom op (autograd, inter-layer dropout when training)."""
    if eattr.requires_grad:
        raise NotImplementedError("gradients w.r.t. bond features are not produced by the backward kernels")
    p = float(model.dropout_rate) if train_dropout else 0.0
    br = _eager_bridge()
    if br is not None and torch.is_autocast_enabled("cuda"):           # (as in lba_encoder: the ops' autocast rule, by hand)
        x, eattr = (t.float() if t.dtype in (torch.float16, torch.bfloat16) else t for t in (x, eattr))
    if br is not None:
        dev, N = x.device, x.shape[0]
        csr = _memo_tables(edge_index, N)
        widths = list(model._widths)
        masks, draw = [], False
        if p > 0 and save_state and len(widths) > 2:
            if PINNED_MASKS is not None:
                masks = [PINNED_MASKS(1, N, widths[l + 1], p, dev)[0].contiguous() for l in range(len(widths) - 2)]
            else:
                draw = True
        params = gine_params(model, one_leaf_ok=True)
        with torch.cuda.device(dev):
            out, ws = br.gine_encoder(params, x, ntypes, eattr, etypes, edge_index, csr, widths,
                                      model.num_ntypes if model._onehot_ntypes else 0,
                                      model.num_etypes, float(slope), p, save_state, masks,
                                      rng_state("gine", dev) if draw else None, None if csr else ops.csr_counters(dev, N),
                                      0, GINE_BWD_WORKGROUPS)
        if draw:
            _LAST_WS["gine"] = (ws, 0)
        return out
    # custom-op path (torch.compile; CGVP_BRIDGE=0): the kernel weights as ONE tensor (see lba_encoder)
    params = gine_params(model)
    flat = params[0] if len(params) == 1 else torch.cat([q.reshape(-1) for q in params])
    out, _, _ = torch.ops.caster_gvp.gine_encoder(
        [flat], x, ntypes, eattr, etypes, edge_index, _memo_tables(edge_index, x.shape[0]), list(model._widths),
        model.num_ntypes if model._onehot_ntypes else 0, model.num_etypes, float(slope), p, save_state)
    return out
