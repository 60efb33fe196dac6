"""The encoders as `torch.library` custom ops (namespace ``caster_gvp``).

Why custom ops and not a bare ``torch.autograd.Function`` around ctypes calls: the reference wraps its model in
``torch.compile(model, dynamic=True)`` (train_model.py:422), runs the forward under ``torch.autocast``
(:561) and back-propagates a ``GradScaler``-scaled loss (:478, :570-587).  With the registrations below all three
work unchanged:

  * ``register_fake``      -- shape/dtype propagation for Dynamo / AOTAutograd (dynamic N, E);
  * ``register_autograd``  -- the backward is itself a custom op (``*_backward``), so AOTAutograd can trace it;
  * ``register_autocast``  -- float inputs are cast to fp32 (the kernels are fp32 storage / fp32 accumulate);
  * the real kernels are opaque to the compiler: each op is the C-ABI launch sequence of one encoder pass.

Ops
  caster_gvp::lba_encoder            VectorProteinGNN_LBAModel.forward (protein_gnn.py:361-388)
  caster_gvp::lba_encoder_backward   what autograd derives from it in the reference
  caster_gvp::gine_encoder           HomoMoleculeGNN_GINE.forward (molecule_gnn.py:254-268)
  caster_gvp::gine_encoder_backward

The forward ops save each stage's INPUTS (node rows h_l, aggregated messages dh_l) and the dropout SEED as extra
outputs; the backward ops launch the hand-written backward kernels, which recompute their stage (regenerating the
same dropout factors from the seed) and emit data gradients plus an arena of weight gradients.  Dropout factors are
generated inside the kernels (Philox keyed by (seed, offset, mask id, node, channel), csrc/gvp_rng.h): the only
PyTorch launch of an encoder pass is the one `randint` that draws {seed, offset} from torch's CUDA generator
(so `torch.manual_seed` governs it and HIP-graph replays advance it).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import torch
from torch import Tensor

from . import _lib, ops
from .ops import EROW, ROW, _act, _f32, _i64, _ptr, _stream

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
# generator the first time it is needed -- and again whenever `torch.manual_seed` / `torch.cuda.manual_seed` has
# changed the generator's seed since -- and from then on every training pass advances `offset` ON THE DEVICE (inside
# the pass's first kernel for the protein encoder, by a 1-thread launch for the drug encoder), leaving the pass's own
# pair in a small output tensor that its later kernels and its backward read.  No torch RNG launch per step, and a
# HIP-graph replay advances the state like an eager step does.
_RNG_STATE = {}
_LAST_SEED = {}


def rng_state(kind, device):
    key = (kind, device.index)
    gen_seed = torch.cuda.default_generators[device.index].initial_seed()
    hit = _RNG_STATE.get(key)
    if hit is None or hit[0] != gen_seed:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the dropout generator state must exist before HIP-graph capture: run one training "
                               "step eagerly first (any warm-up does)")
        hit = (gen_seed, draw_seed(device))
        _RNG_STATE[key] = hit
    return hit[1]


def last_seed(kind):
    """The {seed, offset} pair of the latest training pass of `kind` ('lba' / 'gine') -- tests export the factors
    the kernels used from it (ops.dropout_masks)."""
    return _LAST_SEED[kind]


def make_cfg(dims, num_ntypes, num_etypes, num_convs, aggr_mean):
    return [int(getattr(dims, f)) for f in _CFG_FIELDS] + [int(num_ntypes), int(num_etypes), int(num_convs),
                                                           1 if aggr_mean else 0]


def _dims_layout(cfg, sdt=torch.float32):
    dims = ops.make_dims(storage=ops.BF16 if sdt == torch.bfloat16 else ops.F32,
                         **{f: cfg[i] for i, f in enumerate(_CFG_FIELDS)})
    return dims, ops.lba_layout(dims, cfg[CFG_NTN], cfg[CFG_NTE], cfg[CFG_NC])


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


_IMAGES = {}        # (arena data_ptr, storage type) -> (weakrefs of the first / last parameter, sum of versions, image)


def invalidate_images():
    """Drop every cached fragment image (needed only after writing weights through `.data`, which bypasses the version
    counters the cache watches)."""
    _IMAGES.clear()


_CAPTURE_IMAGE = {}     # key -> image built by the latest FORWARD while a stream capture was in progress


_IMAGE_FLOATS = {}


def image_floats(layout, dims):
    key = (layout.nt_node, layout.nt_edge, layout.total, int(dims.storage))
    n = _IMAGE_FLOATS.get(key)
    if n is None:
        n = _IMAGE_FLOATS[key] = int(_lib.lib().cgvp_lba_image_floats(C.byref(dims), C.byref(layout)))
    return n


def remember_image(params, flat, dims, image):
    """Register an image another launch built (cgvp_lba_pass_begin) for the backward of the same pass / later eager calls."""
    import weakref
    key = (flat.data_ptr(), int(dims.storage))
    if torch.cuda.is_current_stream_capturing():
        _CAPTURE_IMAGE.clear()
        _CAPTURE_IMAGE[key] = image
        return
    if len(_IMAGES) > 16:
        _IMAGES.clear()
    try:
        _IMAGES[key] = (weakref.ref(params[0]), weakref.ref(params[-1]), sum(p._version for p in params), image)
    except TypeError:
        pass


def fragment_image(params, flat, layout, dims, backward=False):
    """Fragment image of the current weights (one ~5 us launch), cached per parameter set: a hit needs the SAME
    parameter tensor objects (weak references -- a new model whose arena lands on a freed model's address is a miss)
    at the same sum of version counters (every in-place update, i.e. every optimizer step, raises it)."""
    import weakref
    key = (flat.data_ptr(), int(dims.storage))
    if torch.cuda.is_current_stream_capturing():
        # a captured step is replayed after the optimizer has changed the weights: the image build must be IN the graph
        # (once per pass: the backward of the captured step reuses the node its forward recorded)
        if backward and key in _CAPTURE_IMAGE:
            return _CAPTURE_IMAGE[key]
        image = ops.prepare_image(flat, layout, dims)
        _CAPTURE_IMAGE.clear()
        _CAPTURE_IMAGE[key] = image
        return image
    _CAPTURE_IMAGE.clear()
    ver = sum(p._version for p in params)
    hit = _IMAGES.get(key)
    if hit is not None and hit[0]() is params[0] and hit[1]() is params[-1] and hit[2] == ver and hit[3].device == flat.device:
        return hit[3]
    image = ops.prepare_image(flat, layout, dims)
    if len(_IMAGES) > 16:
        _IMAGES.clear()
    try:
        _IMAGES[key] = (weakref.ref(params[0]), weakref.ref(params[-1]), ver, image)
    except TypeError:                      # not weak-referenceable (e.g. a traced stand-in): do not cache
        pass
    return image


# ===================================================================================== protein encoder
@torch.library.custom_op("caster_gvp::lba_encoder", mutates_args=(), device_types="cuda")
def lba_encoder_op(params: List[Tensor], x_s: Tensor, x_v: Tensor, ntypes: Tensor, e_s: Tensor, e_v: Tensor,
                   etypes: Tensor, edge_index: Tensor, cfg: List[int], dropout_p: float,
                   save_state: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> (out [N, out_s], state [2 L + 1, N, 28] = h_0..h_{L-1}, dh_0..dh_{L-1}, head input,
           masks (empty unless the PINNED_MASKS test hook is set: [2 L, N, 20]), seed int64[2] (empty without dropout),
           e_emb [E, 36]: the edge embedding store in sorted-edge order (layer 0 writes it, later layers and the
           backward read it)).
    save_state=False is the inference launch sequence (state / masks / seed / e_emb come back empty)."""
    if ops.VARIANT != "mfma" and save_state:
        raise NotImplementedError("training / gradients need the MFMA kernels (CGVP_VARIANT=mfma)")
    L = _lib.lib()
    sdt = ops.storage_dtype(x_s)                       # activation storage: bf16 when the features arrive in bf16
    dims, layout = _dims_layout(cfg, sdt)
    nc, mean = cfg[CFG_NC], bool(cfg[CFG_MEAN])
    x_s, x_v, e_s, e_v = _act(x_s, "x_s", sdt), _act(x_v, "x_v", sdt), _act(e_s, "eattr_s", sdt), _act(e_v, "eattr_v", sdt)
    N, E = int(x_s.shape[0]), int(edge_index.shape[1])      # e_s / e_v may be a resident feature table (CsrStore table mode)
    dev = x_s.device
    flat = flat_arena(params)
    if flat.numel() != layout.total:
        raise RuntimeError(f"parameter arena has {flat.numel()} floats, kernels expect {layout.total}")
    f32, act = dict(dtype=torch.float32, device=dev), dict(dtype=sdt, device=dev)
    if not save_state:
        image, h0 = None, None
        if ops.VARIANT == "mfma" and tuple(x_s.shape) == (N, dims.node_in_s) and tuple(x_v.shape) == (N, dims.node_in_v, 3):
            # inference: the same first launch as a training pass (node embedding + image + edge counts), no generator
            image = torch.empty(image_floats(layout, dims), **f32)
            h0 = torch.empty(N, ROW, **act)
            if edge_index.dim() != 2 or edge_index.shape[0] != 2:
                raise ValueError(f"edge_index must be [2, E], got {tuple(edge_index.shape)}")
            memo = ops.csr_memo(edge_index, N)
            counters = ops.csr_counters(dev, N) if (memo is None and E > 0) else None
            ei = ops._i64(edge_index, "edge_index") if counters is not None else None
            nt0 = _i64(ntypes, "ntypes") if layout.nt_node > 0 else None
            with torch.cuda.device(dev):
                _lib.check(L.cgvp_lba_pass_begin(C.byref(dims), C.byref(layout), _ptr(flat), _ptr(image), _ptr(x_s), _ptr(x_v),
                                                 _ptr(nt0), N, _ptr(h0), None, None, _ptr(ei), E, _ptr(counters), _stream()),
                           "cgvp_lba_pass_begin")
            csr = ops.csr_for_forward(edge_index, N, counted=counters)
        else:
            csr = ops.csr_for_forward(edge_index, N)
        out = ops.lba_encoder_forward(flat, layout, dims, nc, x_s, x_v, ntypes, e_s, e_v, etypes, csr,
                                      aggr_mean=mean, image=image, h0=h0)
        return (out, torch.empty(0, **act), torch.empty(0, **f32), torch.empty(0, dtype=torch.int64, device=dev),
                torch.empty(0, **act))
    if tuple(x_s.shape) != (N, dims.node_in_s) or tuple(x_v.shape) != (N, dims.node_in_v, 3):
        raise NotImplementedError("feature shapes do not match the compiled CASTER-DTA configuration")
    nt = _i64(ntypes, "ntypes") if layout.nt_node > 0 else None
    et = _i64(etypes, "etypes") if layout.nt_edge > 0 else None
    state = torch.empty(2 * nc + 1, N + (N & 1), ROW, **act)      # even row count: every slice stays 16-B aligned in bf16 too
    if N & 1:
        state[:, N].zero_()                                       # the pad row is an op output: keep it deterministic
    hs, dhs, h_last = [state[l] for l in range(nc)], [state[nc + l] for l in range(nc)], state[2 * nc]
    masks, mk, seed = torch.empty(0, **f32), [(None, None)] * nc, torch.empty(0, dtype=torch.int64, device=dev)
    if dropout_p > 0 and PINNED_MASKS is not None:
        masks = PINNED_MASKS(2 * nc, N, MROW, dropout_p, dev)
        mk = [(masks[2 * l], masks[2 * l + 1]) for l in range(nc)]
    elif dropout_p > 0:
        seed = torch.empty(2, dtype=torch.int64, device=dev)      # filled by the node-embed kernel (below)
        _LAST_SEED["lba"] = seed
    rstate = rng_state("lba", dev) if seed.numel() else None
    out = torch.empty(N, dims.out_s, **act)
    e_emb = torch.empty(E + 1, EROW, **act)
    with torch.cuda.device(dev):
        d, lay, P = C.byref(dims), C.byref(layout), _ptr(flat)
        # ONE launch for the three independent things a pass starts with: the node embedding, the fragment image of the
        # current weights and the edge counts of the CSR build (each is launch latency at these sizes).
        # (Forking them onto a second stream instead was measured in round 2: every cross-queue join costs ~10 us.)
        st = _stream()
        image = torch.empty(image_floats(layout, dims), **f32)
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError(f"edge_index must be [2, E], got {tuple(edge_index.shape)}")
        memo = ops.csr_memo(edge_index, N)
        counters = ops.csr_counters(dev, N) if (memo is None and E > 0) else None
        ei = ops._i64(edge_index, "edge_index") if counters is not None else None
        _lib.check(L.cgvp_lba_pass_begin(d, lay, P, _ptr(image), _ptr(x_s), _ptr(x_v), _ptr(nt), N, _ptr(hs[0]),
                                         _ptr(rstate), _ptr(seed if rstate is not None else None),
                                         _ptr(ei), E, _ptr(counters), st), "cgvp_lba_pass_begin")
        remember_image(params, flat, dims, image)
        I = _ptr(image)
        csr = ops.csr_for_forward(edge_index, N, counted=counters)
        rows = csr.table_rows if csr.table_rows is not None else E
        if tuple(e_s.shape) != (rows, dims.edge_in_s) or tuple(e_v.shape) != (rows, dims.edge_in_v, 3) or csr.num_edges != E:
            raise NotImplementedError("feature shapes do not match the compiled CASTER-DTA configuration")
        for l in range(nc):
            last = l == nc - 1
            rng = ops.make_rng(seed, dropout_p, 2 * l)
            e_in, e_out = _ptr(e_emb if l > 0 else None), _ptr(e_emb if l == 0 else None)
            if ops.fuse_layer(N, E):
                with ops._timed("conv_fwd"):
                    _lib.check(L.cgvp_conv_layer_fwd(d, lay, I, l, _ptr(hs[l]), _ptr(e_s), _ptr(e_v), _ptr(et),
                                                     _ptr(csr.rowptr), _ptr(csr.eperm), _ptr(csr.esrc),
                                                     _ptr(csr.edst), N, E, 1 if mean else 0,
                                                     _ptr(mk[l][0]), _ptr(mk[l][1]), ops._rng_ref(rng),
                                                     1 if last else 0, e_in, e_out, _ptr(dhs[l]), _ptr(h_last if last else hs[l + 1]), _ptr(out),
                                                     st), "cgvp_conv_layer_fwd")
                continue
            with ops._timed("conv_fwd"):
                _lib.check(L.cgvp_conv_fwd(d, lay, P, I, l, _ptr(hs[l]), _ptr(e_s), _ptr(e_v), _ptr(et),
                                           _ptr(csr.rowptr), _ptr(csr.eperm), _ptr(csr.esrc), _ptr(csr.edst),
                                           N, E, 1 if mean else 0, e_in, e_out, _ptr(dhs[l]), st), "cgvp_conv_fwd")
            _lib.check(L.cgvp_node_update_fwd_train(d, lay, I, l, _ptr(hs[l]), _ptr(dhs[l]), _ptr(mk[l][0]),
                                                    _ptr(mk[l][1]), ops._rng_ref(rng), N, 1 if last else 0,
                                                    _ptr(h_last if last else hs[l + 1]), _ptr(out), st),
                       "cgvp_node_update_fwd_train")
    return out, state, masks, seed, e_emb


@lba_encoder_op.register_fake
def _(params, x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, cfg, dropout_p, save_state):
    N, nc = x_s.shape[0], cfg[CFG_NC]
    sdt = torch.bfloat16 if x_s.dtype == torch.bfloat16 else torch.float32
    out = x_s.new_empty((N, cfg[8]), dtype=sdt)
    e32, e64 = x_s.new_empty((0,), dtype=torch.float32), x_s.new_empty((0,), dtype=torch.int64)
    if not save_state:
        return out, x_s.new_empty((0,), dtype=sdt), e32, e64, x_s.new_empty((0,), dtype=sdt)
    pinned = dropout_p > 0 and PINNED_MASKS is not None
    masks = x_s.new_empty((2 * nc, N, MROW), dtype=torch.float32) if pinned else e32
    seed = x_s.new_empty((2,), dtype=torch.int64) if dropout_p > 0 and not pinned else e64
    e_emb = x_s.new_empty((edge_index.shape[1] + 1, EROW), dtype=sdt)
    return out, x_s.new_empty((2 * nc + 1, (N + 1) // 2 * 2, ROW), dtype=sdt), masks, seed, e_emb


@torch.library.custom_op("caster_gvp::lba_encoder_backward", mutates_args=(), device_types="cuda")
def lba_encoder_backward_op(g_out: Tensor, params: List[Tensor], x_s: Tensor, x_v: Tensor, ntypes: Tensor,
                            e_s: Tensor, e_v: Tensor, etypes: Tensor, edge_index: Tensor, state: Tensor,
                            masks: Tensor, seed: Tensor, e_emb: Tensor, cfg: List[int], dropout_p: float,
                            need_x: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (grad arena [layout.total], g_x_s [N, 17], g_x_v [N, 3, 3]) (the latter two empty unless need_x)."""
    L = _lib.lib()
    sdt = ops.storage_dtype(x_s)
    dims, layout = _dims_layout(cfg, sdt)
    nc, mean = cfg[CFG_NC], bool(cfg[CFG_MEAN])
    x_s, x_v, e_s, e_v = _act(x_s, "x_s", sdt), _act(x_v, "x_v", sdt), _act(e_s, "eattr_s", sdt), _act(e_v, "eattr_v", sdt)
    state, e_emb = _act(state, "state", sdt), _act(e_emb, "e_emb", sdt)
    N, E = int(x_s.shape[0]), int(edge_index.shape[1])
    dev = x_s.device
    nt = _i64(ntypes, "ntypes") if layout.nt_node > 0 else None
    et = _i64(etypes, "etypes") if layout.nt_edge > 0 else None
    flat = flat_arena(params)
    image = fragment_image(params, flat, layout, dims, backward=True)
    csr = ops.csr_for_backward(edge_index, N)
    hs, dhs, h_last = [state[l] for l in range(nc)], [state[nc + l] for l in range(nc)], state[2 * nc]
    mk = [(masks[2 * l], masks[2 * l + 1]) for l in range(nc)] if masks.numel() else [(None, None)] * nc
    g_out = _f32(g_out.float(), "grad_output")               # every gradient buffer is fp32, whatever the storage type
    f32 = dict(dtype=torch.float32, device=dev)
    gparams = torch.empty(layout.total, **f32)        # every element is STORED by the final reduce (disjoint segments)
    # every stage writes its per-workgroup partial weight-gradient blocks into its own region of
    # one workspace; a single reduce launch at the end sums them all in a fixed order
    wsz = int(L.cgvp_bwd_workspace_floats(C.byref(dims), C.byref(layout)))
    nstage = 2 * nc + 2
    ws_all = torch.empty(nstage * wsz, **f32)
    g_e = torch.empty(nc, E + 1, EROW, **f32)         # d(edge embedding) of every conv layer, summed by the edge stage
    segs = (_lib.Segment * (2 * nstage))()
    nseg, stage = 0, 0
    cnt = C.c_int32(0)

    def region():
        nonlocal stage
        r = ws_all[stage * wsz:(stage + 1) * wsz]
        stage += 1
        return r

    def take():
        nonlocal nseg
        nseg += cnt.value
    g_x_s = torch.empty(N, dims.node_in_s, **f32) if need_x else None
    g_x_v = torch.empty(N, dims.node_in_v, 3, **f32) if need_x else None
    with torch.cuda.device(dev):
        st = _stream()
        d, lay, I = C.byref(dims), C.byref(layout), _ptr(image)
        ups = (None, None, None)        # gradient w.r.t. the output of layer l (sum of up to 3 buffers)
        for l in reversed(range(nc)):
            last = l == nc - 1
            g_dh = torch.empty(N, ROW, **f32)
            g_h = torch.empty(N, ROW, **f32) if (mk[l][0] is not None or seed.numel()) else None
            g_src = torch.empty(N, ROW, **f32)       # zeroed by the node stage, filled by the conv stage's atomics
            rng = ops.make_rng(seed, dropout_p, 2 * l)
            _lib.check(L.cgvp_node_update_bwd(d, lay, I, l, _ptr(hs[l]), _ptr(dhs[l]), _ptr(mk[l][0]),
                                              _ptr(mk[l][1]), ops._rng_ref(rng), _ptr(h_last if last else None),
                                              _ptr(g_out if last else None), _ptr(ups[0]),
                                              _ptr(ups[1]), _ptr(ups[2]), N, 1 if last else 0, _ptr(g_dh),
                                              _ptr(g_h), _ptr(g_src), _ptr(gparams), _ptr(region()),
                                              C.byref(segs, nseg * C.sizeof(_lib.Segment)), C.byref(cnt), st),
                       "cgvp_node_update_bwd")
            take()
            g_dst = torch.empty(N, ROW, **f32)
            with ops._timed("conv_bwd"):
                _lib.check(L.cgvp_conv_bwd(d, lay, I, l, _ptr(hs[l]), _ptr(e_emb),
                                           _ptr(csr.rowptr), _ptr(csr.esrc), _ptr(csr.edst), N, E,
                                           1 if mean else 0, _ptr(g_dh), _ptr(g_src), 1, _ptr(g_dst), _ptr(g_e[l]),
                                           _ptr(gparams), _ptr(region()),
                                           C.byref(segs, nseg * C.sizeof(_lib.Segment)), C.byref(cnt), st),
                           "cgvp_conv_bwd")
            take()
            ups = (g_h if g_h is not None else g_dh, g_src, g_dst)
        _lib.check(L.cgvp_node_embed_bwd(d, lay, I, _ptr(x_s), _ptr(x_v), _ptr(nt), N, _ptr(ups[0]), _ptr(ups[1]),
                                         _ptr(ups[2]), _ptr(g_x_s), _ptr(g_x_v), _ptr(gparams), _ptr(region()),
                                         C.byref(segs, nseg * C.sizeof(_lib.Segment)), C.byref(cnt), st),
                   "cgvp_node_embed_bwd")
        take()
        if E > 0:
            ge = (C.c_void_p * nc)(*[g_e[l].data_ptr() for l in range(nc)])
            _lib.check(L.cgvp_edge_embed_bwd(d, lay, I, _ptr(e_s), _ptr(e_v), _ptr(et), _ptr(csr.eperm), E, ge, nc,
                                             _ptr(gparams), _ptr(region()),
                                             C.byref(segs, nseg * C.sizeof(_lib.Segment)), C.byref(cnt), st),
                       "cgvp_edge_embed_bwd")
            take()
        if E == 0:                                    # no edge stage: gvp_edge's gradient block is not covered by a segment
            gparams[layout.edge_gvp:layout.conv0].zero_()
        _lib.check(L.cgvp_bwd_reduce(segs, nseg, _ptr(gparams), 1, st), "cgvp_bwd_reduce")
    return gparams, (g_x_s if need_x else torch.empty(0, **f32)), (g_x_v if need_x else torch.empty(0, **f32))


@lba_encoder_backward_op.register_fake
def _(g_out, params, x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, state, masks, seed, e_emb, cfg, dropout_p, need_x):
    total = sum(p.numel() for p in params)
    gp = x_s.new_empty((total,), dtype=torch.float32)
    if need_x:
        return gp, x_s.new_empty(x_s.shape, dtype=torch.float32), x_v.new_empty(x_v.shape, dtype=torch.float32)
    return gp, x_s.new_empty((0,), dtype=torch.float32), x_s.new_empty((0,), dtype=torch.float32)


def _lba_setup(ctx, inputs, output):
    params, x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, cfg, dropout_p, save_state = inputs
    out, state, masks, seed, e_emb = output
    if not save_state:
        raise RuntimeError("caster_gvp::lba_encoder was run with save_state=False; gradients need save_state=True")
    ctx.cfg, ctx.dropout_p = cfg, dropout_p
    ctx.shapes = [tuple(p.shape) for p in params]
    ctx.set_materialize_grads(False)      # state / e_emb are outputs only to be saved: no zero-filled "gradients" for them
    ctx.save_for_backward(x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, state, masks, seed, e_emb, *params)


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


def _lba_backward(ctx, g_out, g_state, g_masks, g_seed, g_e_emb):
    if g_out is None:
        return (None,) * 11
    x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, state, masks, seed, e_emb, *params = ctx.saved_tensors
    need_x = bool(ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
    gflat, g_x_s, g_x_v = torch.ops.caster_gvp.lba_encoder_backward(
        g_out.contiguous(), params, x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, state, masks, seed, e_emb, ctx.cfg,
        ctx.dropout_p, need_x)
    grads = [g.view(s) for g, s in zip(torch.split(gflat, [_numel(s) for s in ctx.shapes]), ctx.shapes)]
    if need_x and x_s.dtype != torch.float32:                 # input gradients are produced in fp32
        g_x_s, g_x_v = g_x_s.to(x_s.dtype), g_x_v.to(x_v.dtype)
    return (grads, g_x_s if need_x else None, g_x_v if need_x else None, None, None, None, None, None, None, None,
            None)


torch.library.register_autograd("caster_gvp::lba_encoder", _lba_backward, setup_context=_lba_setup)
torch.library.register_autocast("caster_gvp::lba_encoder", "cuda", torch.float32)


def lba_encoder(model, x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, train_dropout, save_state):
    """VectorProteinGNN_LBAModel.forward through the custom op (autograd, dropout when training)."""
    if e_s.requires_grad or e_v.requires_grad:
        raise NotImplementedError("gradients w.r.t. raw edge features are not produced by the backward kernels")
    # plain Python ints only (no ctypes objects here: this function is traced by Dynamo under torch.compile);
    # the op validates them against the compiled kernel configuration
    cfg = [model.in_channels[0], model.in_channels[1], model.edge_dim[0], model.edge_dim[1],
           model.hidden_channels[0], model.hidden_channels[1], model.edge_hidden_channels[0],
           model.edge_hidden_channels[1], model.out_channels[0], model.num_ntypes, model.num_etypes,
           model.num_convs, 1 if model.aggr == "mean" else 0]
    out, _, _, _, _ = torch.ops.caster_gvp.lba_encoder(
        model.op_params(), x_s, x_v, ntypes, e_s, e_v, etypes, edge_index, cfg,
        float(model.dropout_rate) if train_dropout else 0.0, save_state)
    return out


# ===================================================================================== drug encoder
_GINE_KEYS = ("eps", "w0", "b0", "w1", "b1", "we", "be")     # slab / state_dict order of one GINEConv


@torch.library.custom_op("caster_gvp::gine_encoder", mutates_args=(), device_types="cuda")
def gine_encoder_op(params: List[Tensor], x: Tensor, ntypes: Tensor, eattr: Tensor, etypes: Tensor,
                    edge_index: Tensor, widths: List[int], num_ntypes: int, num_etypes: int, slope: float,
                    dropout_p: float, save_state: bool) -> Tuple[Tensor, List[Tensor], List[Tensor], Tensor]:
    """-> (out [N, widths[-1]], hidden [h_1 .. h_{L-1}], masks [m_0 .. m_{L-2}] (PINNED_MASKS test hook only; empty
    tensors otherwise), seed int64[2] (empty without dropout)) -- the lists are empty when save_state is False."""
    nl = len(widths) - 1
    x, eattr = _f32(x, "x"), _f32(eattr, "eattr")
    N = int(x.shape[0])
    csr = ops.csr_for_forward(edge_index, N)
    ws = [dict(zip(_GINE_KEYS, params[7 * l:7 * l + 7])) for l in range(nl)]
    hs, masks = [x], []
    seed = x.new_empty(0, dtype=torch.int64)
    if dropout_p > 0 and nl > 1 and PINNED_MASKS is None:
        seed = torch.empty(2, dtype=torch.int64, device=x.device)
        _LAST_SEED["gine"] = seed
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().cgvp_rng_next(_ptr(rng_state("gine", x.device)), _ptr(seed), _stream()), "cgvp_rng_next")
    for l in range(nl):
        mask, rng = None, None
        if dropout_p > 0 and l < nl - 1:
            if PINNED_MASKS is not None:
                mask = PINNED_MASKS(1, N, widths[l + 1], dropout_p, x.device)[0]
            else:
                rng = ops.make_rng(seed, dropout_p, l)
        masks.append(mask)
        first = l == 0
        hs.append(ops.gine_conv_forward(hs[l], ntypes if first else None, num_ntypes if first else 0,
                                        eattr, etypes, num_etypes, csr, ws[l], widths[l],
                                        widths[l + 1], widths[l + 1], slope, mask=mask, rng=rng))
    if not save_state:
        return hs[-1], [], [], x.new_empty(0, dtype=torch.int64)
    return hs[-1], hs[1:-1], [m if m is not None else x.new_empty(0) for m in masks[:-1]], seed


@gine_encoder_op.register_fake
def _(params, x, ntypes, eattr, etypes, edge_index, widths, num_ntypes, num_etypes, slope, dropout_p, save_state):
    N, nl = x.shape[0], len(widths) - 1
    out = x.new_empty((N, widths[-1]), dtype=torch.float32)
    e64 = x.new_empty((0,), dtype=torch.int64)
    if not save_state:
        return out, [], [], e64
    hidden = [x.new_empty((N, widths[l]), dtype=torch.float32) for l in range(1, nl)]
    pinned = dropout_p > 0 and PINNED_MASKS is not None
    masks = [x.new_empty((N, widths[l + 1]) if pinned else (0,), dtype=torch.float32) for l in range(nl - 1)]
    seed = x.new_empty((2,), dtype=torch.int64) if dropout_p > 0 and nl > 1 and not pinned else e64
    return out, hidden, masks, seed


@torch.library.custom_op("caster_gvp::gine_encoder_backward", mutates_args=(), device_types="cuda")
def gine_encoder_backward_op(g_out: Tensor, params: List[Tensor], x: Tensor, ntypes: Tensor, eattr: Tensor,
                             etypes: Tensor, edge_index: Tensor, hidden: List[Tensor], masks: List[Tensor],
                             seed: Tensor, widths: List[int], num_ntypes: int, num_etypes: int, slope: float,
                             dropout_p: float, need_x: bool, bwd_workgroups: int) -> Tuple[Tensor, Tensor]:
    """-> (the gradients of `params`, flattened and concatenated in that order, g_x (empty unless need_x))."""
    L = _lib.lib()
    nl = len(widths) - 1
    x, eattr = _f32(x, "x"), _f32(eattr, "eattr")
    N, dev = int(x.shape[0]), x.device
    csr = ops.csr_for_backward(edge_index, N)
    f32 = dict(dtype=torch.float32, device=dev)
    nt, et = _i64(ntypes, "ntypes"), _i64(etypes, "etypes")
    hs = [x] + list(hidden)
    mk = [masks[l] if l < len(masks) and masks[l].numel() else None for l in range(nl)]
    ws = [dict(zip(_GINE_KEYS, params[7 * l:7 * l + 7])) for l in range(nl)]
    g = _f32(g_out, "grad_output")
    wsp = torch.empty(int(L.cgvp_gine_bwd_workspace_floats()), **f32)
    layer_floats = [sum(p.numel() for p in params[7 * l:7 * l + 7]) for l in range(nl)]
    gflat = torch.zeros(sum(layer_floats), **f32)
    edge_dim = int(eattr.shape[1])
    with torch.cuda.device(dev):
        for l in reversed(range(nl)):
            first = l == 0
            cin, cout = widths[l], widths[l + 1]
            w = {k: _f32(v, k) for k, v in ws[l].items()}
            glayer = gflat[sum(layer_floats[:l]):sum(layer_floats[:l + 1])]
            want_x = (not first) or need_x
            g_x = torch.empty(N, cin - (num_ntypes if first else 0), **f32) if want_x else None
            gw = _lib.GineW(**{k: v.data_ptr() for k, v in w.items()})
            rng = ops.make_rng(seed, dropout_p, l) if l < nl - 1 else None
            rc = L.cgvp_gine_conv_bwd(_ptr(hs[l]), _ptr(nt if first else None), num_ntypes if first else 0,
                                      _ptr(eattr), _ptr(et), num_etypes, edge_dim, _ptr(csr.rowptr),
                                      _ptr(csr.eperm), _ptr(csr.esrc), _ptr(csr.edst), N, csr.num_edges, cin, cout, cout,
                                      C.byref(gw), float(slope), _ptr(mk[l]), ops._rng_ref(rng), _ptr(g), _ptr(g_x),
                                      _ptr(glayer), _ptr(wsp), int(bwd_workgroups), _stream())
            _lib.check(rc, "cgvp_gine_conv_bwd")
            g = g_x
    return gflat, (g if need_x else torch.empty(0, **f32))


@gine_encoder_backward_op.register_fake
def _(g_out, params, x, ntypes, eattr, etypes, edge_index, hidden, masks, seed, widths, num_ntypes, num_etypes, slope,
      dropout_p, need_x, bwd_workgroups):
    gflat = x.new_empty((sum(p.numel() for p in params),), dtype=torch.float32)
    return gflat, (x.new_empty(x.shape, dtype=torch.float32) if need_x else x.new_empty((0,), dtype=torch.float32))


def _gine_setup(ctx, inputs, output):
    params, x, ntypes, eattr, etypes, edge_index, widths, num_ntypes, num_etypes, slope, dropout_p, save_state = inputs
    out, hidden, masks, seed = output
    if not save_state:
        raise RuntimeError("caster_gvp::gine_encoder was run with save_state=False; gradients need save_state=True")
    ctx.meta = (widths, num_ntypes, num_etypes, slope, dropout_p, len(params), len(hidden))
    ctx.shapes = [tuple(p.shape) for p in params]
    ctx.set_materialize_grads(False)
    ctx.save_for_backward(x, ntypes, eattr, etypes, edge_index, seed, *params, *hidden, *masks)


def _gine_backward(ctx, g_out, g_hidden, g_masks, g_seed):
    if g_out is None:
        return (None,) * 12
    widths, num_ntypes, num_etypes, slope, dropout_p, n_params, n_hidden = ctx.meta
    x, ntypes, eattr, etypes, edge_index, seed, *rest = ctx.saved_tensors
    params, hidden, masks = rest[:n_params], rest[n_params:n_params + n_hidden], rest[n_params + n_hidden:]
    need_x = bool(ctx.needs_input_grad[1])
    gflat, g_x = torch.ops.caster_gvp.gine_encoder_backward(
        g_out.contiguous(), params, x, ntypes, eattr, etypes, edge_index, hidden, masks, seed, widths, num_ntypes,
        num_etypes, slope, dropout_p, need_x, GINE_BWD_WORKGROUPS)
    grads = [g.view(s) for g, s in zip(torch.split(gflat, [_numel(s) for s in ctx.shapes]), ctx.shapes)]
    return (grads, g_x if need_x else None, None, None, None, None, None, None, None, None, None, None)


# Cap on the CUs of the GINE backward (0 = the library default of 16: inside JointGNN it runs beside the
# protein backward, whose kernels own 240 of the 256 CUs).  A host that trains the drug encoder alone may
# raise it; it is forwarded as the `max_workgroups` ARGUMENT of cgvp_gine_conv_bwd (the library has no state).
GINE_BWD_WORKGROUPS = 0

torch.library.register_autograd("caster_gvp::gine_encoder", _gine_backward, setup_context=_gine_setup)
torch.library.register_autocast("caster_gvp::gine_encoder", "cuda", torch.float32)


def gine_encoder(model, x, ntypes, eattr, etypes, edge_index, slope, train_dropout, save_state):
    """HomoMoleculeGNN_GINE.forward through the custom op (autograd, inter-layer dropout when training)."""
    if eattr.requires_grad:
        raise NotImplementedError("gradients w.r.t. bond features are not produced by the backward kernels")
    params = []
    for conv in model.conv_list:
        kw = conv.kernel_weights()
        params += [kw[k] for k in _GINE_KEYS]
    out, _, _, _ = torch.ops.caster_gvp.gine_encoder(
        params, x, ntypes, eattr, etypes, edge_index, list(model._widths), model.num_ntypes, model.num_etypes,
        float(slope), float(model.dropout_rate) if train_dropout else 0.0, save_state)
    return out
