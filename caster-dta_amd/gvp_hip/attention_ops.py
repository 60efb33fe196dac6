"""Varlen residue <-> atom cross attention as `torch.library` custom ops.

`caster_gvp::cross_attention` is the part of the two nn.MultiheadAttention modules of CrossAttentionModule
(joint_gnn.py:321-409) that lies between their input and output projections, for BOTH directions in one launch, on
compact row arrays with ptr offsets (no to_dense_batch padding, no key-padding masks, no score tensor in HBM):

    o_r[q] = softmax_k(q_r[q] . k_a[k] / sqrt(16)) v_a[k]      residues attend to the atoms of their pair
    o_a[q] = softmax_k(q_a[q] . k_r[k] / sqrt(16)) v_r[k]      atoms attend to the residues of their pair

The projections themselves stay plain library GEMMs (F.linear on [N, 128] rows).  Registered with fake kernels,
autograd (the backward is the op `cross_attention_backward`: two launches, no atomics) and an autocast rule (fp32),
like the encoders.  `cross_attention_weights` produces nn.MultiheadAttention's head-averaged weights in the
reference's dense [B, Lq_max, Lk_max] layout for inference (inference/evaluation.py:43-66).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Tuple

import torch
from torch import Tensor

from . import _lib
from .ops import _f32, _ptr, _stream

HEAD_DIM = 16


def _i64c(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: caster-dta_amd runs on MI355X only (got a {t.device} tensor)")
    return t.to(torch.int64).contiguous()


def _problems(q_r, k_a, v_a, q_a, k_r, v_r, rptr, aptr):
    P = (_lib.AttnProblem * 2)()
    for p, (q, k, v, qp, kp) in zip(P, ((q_r, k_a, v_a, rptr, aptr), (q_a, k_r, v_r, aptr, rptr))):
        p.q, p.k, p.v = q.data_ptr(), k.data_ptr(), v.data_ptr()
        p.q_ptr, p.k_ptr = qp.data_ptr(), kp.data_ptr()
        p.num_q, p.num_k = int(q.shape[0]), int(k.shape[0])
    return P


def _check(q_r, k_a, v_a, q_a, k_r, v_r, rptr, aptr, heads):
    E = heads * HEAD_DIM
    for name, t in (("q_r", q_r), ("k_a", k_a), ("v_a", v_a), ("q_a", q_a), ("k_r", k_r), ("v_r", v_r)):
        if t.dim() != 2 or t.shape[1] != E:
            raise NotImplementedError(f"{name}: the attention kernels are compiled for head_dim 16 "
                                      f"(embed = 16 * heads = {E}), got {tuple(t.shape)}")
    if not (q_r.shape[0] == k_r.shape[0] == v_r.shape[0] and q_a.shape[0] == k_a.shape[0] == v_a.shape[0]):
        raise ValueError("row counts of the residue / atom tensors disagree")
    if rptr.shape != aptr.shape or rptr.dim() != 1 or rptr.numel() < 1:
        raise ValueError("rptr / aptr must both be [num_pairs + 1]")


@torch.library.custom_op("caster_gvp::cross_attention", mutates_args=(), device_types="cuda")
def cross_attention_op(q_r: Tensor, k_a: Tensor, v_a: Tensor, q_a: Tensor, k_r: Tensor, v_r: Tensor, rptr: Tensor,
                       aptr: Tensor, heads: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """-> (o_r [N, E], o_a [Na, E], lse_r [N, heads], lse_a [Na, heads])."""
    _check(q_r, k_a, v_a, q_a, k_r, v_r, rptr, aptr, heads)
    ts = [_f32(t, "attention operand") for t in (q_r, k_a, v_a, q_a, k_r, v_r)]
    rptr, aptr = _i64c(rptr, "rptr"), _i64c(aptr, "aptr")
    dev = ts[0].device
    N, Na, B = int(q_r.shape[0]), int(q_a.shape[0]), int(rptr.numel()) - 1
    f32 = dict(dtype=torch.float32, device=dev)
    o_r, o_a = torch.empty(N, heads * HEAD_DIM, **f32), torch.empty(Na, heads * HEAD_DIM, **f32)
    lse_r, lse_a = torch.empty(N, heads, **f32), torch.empty(Na, heads, **f32)
    P = _problems(*ts, rptr, aptr)
    P[0].out, P[0].lse, P[1].out, P[1].lse = _ptr(o_r).value or 0, _ptr(lse_r).value or 0, _ptr(o_a).value or 0, _ptr(lse_a).value or 0
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().cgvp_attn_fwd(P, 2, B, heads, 1.0 / math.sqrt(HEAD_DIM), _stream()), "cgvp_attn_fwd")
    return o_r, o_a, lse_r, lse_a


@cross_attention_op.register_fake
def _(q_r, k_a, v_a, q_a, k_r, v_r, rptr, aptr, heads):
    f = lambda t, *s: t.new_empty(s, dtype=torch.float32)
    return (f(q_r, q_r.shape[0], q_r.shape[1]), f(q_a, q_a.shape[0], q_a.shape[1]), f(q_r, q_r.shape[0], heads),
            f(q_a, q_a.shape[0], heads))


@torch.library.custom_op("caster_gvp::cross_attention_backward", mutates_args=(), device_types="cuda")
def cross_attention_backward_op(g_o_r: Tensor, g_o_a: Tensor, q_r: Tensor, k_a: Tensor, v_a: Tensor, q_a: Tensor,
                                k_r: Tensor, v_r: Tensor, o_r: Tensor, o_a: Tensor, lse_r: Tensor, lse_a: Tensor,
                                rptr: Tensor, aptr: Tensor,
                                heads: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> (g_q_r, g_k_a, g_v_a, g_q_a, g_k_r, g_v_r)."""
    ts = [_f32(t, "attention operand") for t in (q_r, k_a, v_a, q_a, k_r, v_r)]
    g_o_r, g_o_a, o_r, o_a = (_f32(t, "attention gradient") for t in (g_o_r, g_o_a, o_r, o_a))
    lse_r, lse_a = _f32(lse_r, "lse"), _f32(lse_a, "lse")
    rptr, aptr = _i64c(rptr, "rptr"), _i64c(aptr, "aptr")
    dev = ts[0].device
    B = int(rptr.numel()) - 1
    outs = [torch.empty_like(t) for t in ts]                 # g_q_r, g_k_a, g_v_a, g_q_a, g_k_r, g_v_r
    d_r, d_a = torch.empty_like(lse_r), torch.empty_like(lse_a)
    P = _problems(*ts, rptr, aptr)
    v = lambda t: _ptr(t).value or 0
    P[0].out, P[0].lse, P[0].g_out, P[0].delta = v(o_r), v(lse_r), v(g_o_r), v(d_r)
    P[0].g_q, P[0].g_k, P[0].g_v = v(outs[0]), v(outs[1]), v(outs[2])
    P[1].out, P[1].lse, P[1].g_out, P[1].delta = v(o_a), v(lse_a), v(g_o_a), v(d_a)
    P[1].g_q, P[1].g_k, P[1].g_v = v(outs[3]), v(outs[4]), v(outs[5])
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().cgvp_attn_bwd(P, 2, B, heads, 1.0 / math.sqrt(HEAD_DIM), _stream()), "cgvp_attn_bwd")
    return tuple(outs)


@cross_attention_backward_op.register_fake
def _(g_o_r, g_o_a, q_r, k_a, v_a, q_a, k_r, v_r, o_r, o_a, lse_r, lse_a, rptr, aptr, heads):
    return tuple(t.new_empty(t.shape, dtype=torch.float32) for t in (q_r, k_a, v_a, q_a, k_r, v_r))


def _attn_setup(ctx, inputs, output):
    q_r, k_a, v_a, q_a, k_r, v_r, rptr, aptr, heads = inputs
    o_r, o_a, lse_r, lse_a = output
    ctx.heads = heads
    ctx.set_materialize_grads(False)          # the log-sum-exp outputs never carry a gradient: no zero fills for them
    ctx.save_for_backward(q_r, k_a, v_a, q_a, k_r, v_r, o_r, o_a, lse_r, lse_a, rptr, aptr)


def _attn_backward(ctx, g_o_r, g_o_a, g_lse_r, g_lse_a):
    q_r, k_a, v_a, q_a, k_r, v_r, o_r, o_a, lse_r, lse_a, rptr, aptr = ctx.saved_tensors
    if g_o_r is None and g_o_a is None:
        return (None,) * 9
    g_o_r = torch.zeros_like(o_r) if g_o_r is None else g_o_r
    g_o_a = torch.zeros_like(o_a) if g_o_a is None else g_o_a
    grads = torch.ops.caster_gvp.cross_attention_backward(g_o_r.contiguous(), g_o_a.contiguous(), q_r, k_a, v_a, q_a,
                                                          k_r, v_r, o_r, o_a, lse_r, lse_a, rptr, aptr, ctx.heads)
    # the two directions share no operand: a residue-side tensor gets its gradient from exactly one of them
    return (*grads, None, None, None)


torch.library.register_autograd("caster_gvp::cross_attention", _attn_backward, setup_context=_attn_setup)
torch.library.register_autocast("caster_gvp::cross_attention", "cuda", torch.float32)


@torch.library.custom_op("caster_gvp::cross_attention_weights", mutates_args=(), device_types="cuda")
def cross_attention_weights_op(q_r: Tensor, k_a: Tensor, lse_r: Tensor, q_a: Tensor, k_r: Tensor, lse_a: Tensor,
                               rptr: Tensor, aptr: Tensor, heads: int, r_max: int,
                               a_max: int) -> Tuple[Tensor, Tensor]:
    """-> (w_r [B, r_max, a_max], w_a [B, a_max, r_max]): head-averaged attention weights, zero in the padding."""
    ts = [_f32(t, "attention operand") for t in (q_r, k_a, k_a, q_a, k_r, k_r)]
    lse_r, lse_a = _f32(lse_r, "lse"), _f32(lse_a, "lse")
    rptr, aptr = _i64c(rptr, "rptr"), _i64c(aptr, "aptr")
    dev = ts[0].device
    B = int(rptr.numel()) - 1
    w_r = torch.zeros(B, r_max, a_max, dtype=torch.float32, device=dev)
    w_a = torch.zeros(B, a_max, r_max, dtype=torch.float32, device=dev)
    dummy_r, dummy_a = torch.empty(1, device=dev), torch.empty(1, device=dev)    # `out` is not written by this entry point
    P = _problems(*ts, rptr, aptr)
    v = lambda t: _ptr(t).value or 0
    P[0].out, P[0].lse, P[0].weights, P[0].weights_lq, P[0].weights_lk = v(dummy_r), v(lse_r), v(w_r), r_max, a_max
    P[1].out, P[1].lse, P[1].weights, P[1].weights_lq, P[1].weights_lk = v(dummy_a), v(lse_a), v(w_a), a_max, r_max
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().cgvp_attn_weights(P, 2, B, heads, 1.0 / math.sqrt(HEAD_DIM), _stream()), "cgvp_attn_weights")
    return w_r, w_a


@cross_attention_weights_op.register_fake
def _(q_r, k_a, lse_r, q_a, k_r, lse_a, rptr, aptr, heads, r_max, a_max):
    B = rptr.shape[0] - 1
    return (q_r.new_empty((B, r_max, a_max), dtype=torch.float32), q_r.new_empty((B, a_max, r_max), dtype=torch.float32))
