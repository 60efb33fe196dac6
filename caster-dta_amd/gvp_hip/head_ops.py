"""Row-wise nn.Linear layers of the joint head with hand-written weight-gradient kernels.

The head applies Linear layers to every residue / atom row of the batch (joint_gnn.py:188-198 residue / atom stacks,
:376-389 attention projections and feed-forward).  Forward and input gradient are ordinary GEMMs (library calls); the
WEIGHT and BIAS gradients reduce over all rows into a 128..384 x 64..256 output, a shape the library serves badly
(44-60 us per layer at 19,200 rows, plus a column-sum kernel per bias; twelve layers per step).  `fast_linear` routes
exactly those through ``caster_gvp::linear_wgrad`` (csrc/linear_kernels.hip: rows split over the chip, fp32 MFMA,
fixed-order reduce) and is `F.linear` everywhere else (small row counts, other dtypes, shapes the kernel is not
compiled for, no gradient wanted).
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import Tensor

from . import _lib
from .ops import _f32, _ptr, _stream

MIN_ROWS = 1024          # below this the library GEMM is latency-bound either way


@torch.library.custom_op("caster_gvp::linear_wgrad", mutates_args=(), device_types="cuda")
def linear_wgrad_op(x: Tensor, gy: Tensor) -> Tensor:
    """x [R, I], gy [R, O] -> flat [O * I + O]: d weight ([O, I] row-major) followed by d bias."""
    L = _lib.lib()
    x, gy = _f32(x, "x"), _f32(gy, "grad_output")
    R, I, O = int(x.shape[0]), int(x.shape[1]), int(gy.shape[1])
    if gy.shape[0] != R:
        raise ValueError("x and grad_output hold different numbers of rows")
    n = int(L.cgvp_linear_wgrad_workspace_floats(R, I, O))
    if n < 0:
        _lib.check(n, "cgvp_linear_wgrad_workspace_floats")
    ws = torch.empty(max(n, 1), dtype=torch.float32, device=x.device)
    out = torch.empty(O * I + O, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.cgvp_linear_wgrad(_ptr(x), _ptr(gy), R, I, O, _ptr(ws), _ptr(out), _stream()), "cgvp_linear_wgrad")
    return out


@linear_wgrad_op.register_fake
def _(x, gy):
    I, O = x.shape[1], gy.shape[1]
    return x.new_empty((O * I + O,), dtype=torch.float32)


def supported(x, weight, bias):
    O, I = weight.shape
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32
            and bias is not None and I % 16 == 0 and 16 <= I <= 256 and O % 64 == 0 and x.shape[0] >= MIN_ROWS
            and torch.is_grad_enabled() and weight.requires_grad and not torch.is_autocast_enabled("cuda"))


class _FastLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        O, I = weight.shape
        flat = torch.ops.caster_gvp.linear_wgrad(x, gy)
        return gx, flat[:O * I].view(O, I), flat[O * I:]


def _bridge():
    """The C++ eager fast path (csrc/torch_bridge.cpp: the same C entry points behind C++ autograd functions, ~60 us of
    host time less per call and direction than a torch.library op), or None: under torch.compile (Dynamo traces the
    custom ops), with `CGVP_BRIDGE=0`, or when the extension is not built."""
    if torch.compiler.is_compiling() or not hasattr(_lib, "bridge"):
        return None
    br = _lib.bridge()
    return br if (br is not None and hasattr(br, "head_linear")) else None


def fast_linear(x, weight, bias):
    """F.linear(x, weight, bias) whose weight / bias gradients come from the split-row kernel when the shape qualifies."""
    if supported(x, weight, bias):
        br = _bridge()
        if br is not None:
            return br.head_linear(x, weight, bias)
        return _FastLinear.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


# ------------------------------------------------------------------------------------------------ row-wise LayerNorm
# nn.LayerNorm on the compact residue / atom rows (joint_gnn.py:376-389).  The stock kernels spend 15 us forward and
# 28 + 12 + 5 us backward on a [19,200 x 128] input (grad-input, partial and final gamma / beta reductions); the library's
# pair (csrc/norm_kernels.hip) is one pass each way: one wave per row, d gamma / d beta accumulated in registers.
LN_MIN_ROWS = 256


@torch.library.custom_op("caster_gvp::layer_norm", mutates_args=(), device_types="cuda")
def layer_norm_op(x: Tensor, weight: Tensor, bias: Tensor, eps: float) -> tuple[Tensor, Tensor, Tensor]:
    """x [R, D] -> (y [R, D], mean [R], rstd [R])."""
    L = _lib.lib()
    x, weight, bias = _f32(x, "x"), _f32(weight, "weight"), _f32(bias, "bias")
    R, D = int(x.shape[0]), int(x.shape[1])
    y = torch.empty_like(x)
    mean = torch.empty(R, dtype=torch.float32, device=x.device)
    rstd = torch.empty(R, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.cgvp_layer_norm_fwd(_ptr(x), _ptr(weight), _ptr(bias), R, D, float(eps), _ptr(y), _ptr(mean), _ptr(rstd),
                                         _stream()), "cgvp_layer_norm_fwd")
    return y, mean, rstd


@layer_norm_op.register_fake
def _(x, weight, bias, eps):
    return torch.empty_like(x), x.new_empty((x.shape[0],)), x.new_empty((x.shape[0],))


@torch.library.custom_op("caster_gvp::layer_norm_backward", mutates_args=(), device_types="cuda")
def layer_norm_backward_op(gy: Tensor, x: Tensor, mean: Tensor, rstd: Tensor, weight: Tensor) -> tuple[Tensor, Tensor]:
    """-> (gx [R, D], [d weight | d bias] flat [2 D])."""
    L = _lib.lib()
    gy, x, weight = _f32(gy, "grad_output"), _f32(x, "x"), _f32(weight, "weight")
    R, D = int(x.shape[0]), int(x.shape[1])
    n = int(L.cgvp_layer_norm_bwd_workspace_floats(R, D))
    if n < 0:
        _lib.check(n, "cgvp_layer_norm_bwd_workspace_floats")
    ws = torch.empty(max(n, 1), dtype=torch.float32, device=x.device)
    gx = torch.empty_like(x)
    gwb = torch.empty(2 * D, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.cgvp_layer_norm_bwd(_ptr(gy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(weight), R, D, _ptr(gx), _ptr(ws),
                                         _ptr(gwb), _stream()), "cgvp_layer_norm_bwd")
    return gx, gwb


@layer_norm_backward_op.register_fake
def _(gy, x, mean, rstd, weight):
    return torch.empty_like(x), x.new_empty((2 * x.shape[1],))


def _ln_setup(ctx, inputs, output):
    x, weight, bias, eps = inputs
    y, mean, rstd = output
    ctx.save_for_backward(x, mean, rstd, weight)
    ctx.set_materialize_grads(False)


def _ln_backward(ctx, gy, g_mean, g_rstd):
    if gy is None:
        return None, None, None, None
    x, mean, rstd, weight = ctx.saved_tensors
    gx, gwb = torch.ops.caster_gvp.layer_norm_backward(gy.contiguous(), x, mean, rstd, weight)
    D = x.shape[1]
    return gx, gwb[:D], gwb[D:], None


torch.library.register_autograd("caster_gvp::layer_norm", _ln_backward, setup_context=_ln_setup)
torch.library.register_autocast("caster_gvp::layer_norm", "cuda", torch.float32)


def fast_layer_norm(x, norm):
    """`norm(x)` for an nn.LayerNorm over the last dim of compact rows [R, D]: the library's kernels when the shape
    qualifies (fp32 CUDA rows, D in {64, 128, 256, 512}, affine), the module itself otherwise."""
    if (isinstance(norm, torch.nn.LayerNorm) and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32
            and norm.elementwise_affine and norm.bias is not None and len(norm.normalized_shape) == 1
            and norm.normalized_shape[0] == x.shape[1] and x.shape[1] in (64, 128, 256, 512)
            and norm.weight.dtype == torch.float32 and x.shape[0] >= LN_MIN_ROWS):
        br = _bridge()
        if br is not None and not torch.is_autocast_enabled("cuda"):
            return br.head_layer_norm(x, norm.weight, norm.bias, float(norm.eps))
        return torch.ops.caster_gvp.layer_norm(x, norm.weight, norm.bias, float(norm.eps))[0]
    return norm(x)


# ------------------------------------------------------------------------------------------------ fused dropout sites
# `x + dropout(a)` and `dropout(act(t))` of the head as ONE launch each way (csrc/elementwise_kernels.hip), masks
# regenerated from a per-step {seed, offset} pair instead of stored.  Eager / captured-graph training in fp32 only: under
# torch.compile Inductor fuses the stock ops itself, under autocast the rows are not fp32 -- both keep the torch ops.
from .ops import Rng

_HEAD_RNG = {}          # device index -> [generator seed it was drawn from, persistent int64[2] {seed, offset}]


def _head_rng_state(device):
    """Persistent device-side generator state of the head's fused dropout sites (never replaced: captured graphs keep
    its address; re-seeded in place after torch.manual_seed)."""
    from . import autograd_ops
    gen_seed = torch.cuda.default_generators[device.index].initial_seed()
    hit = _HEAD_RNG.get(device.index)
    if hit is None or hit[0] != gen_seed:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the dropout generator state must exist before HIP-graph capture: run one training step "
                               "eagerly first (any warm-up does)")
        seed = autograd_ops.draw_seed(device)
        if hit is None:
            _HEAD_RNG[device.index] = hit = [gen_seed, seed]
        else:
            hit[1].copy_(seed)
            hit[0] = gen_seed
    return hit[1]


@torch.library.custom_op("caster_gvp::rng_next", mutates_args=("state",), device_types="cuda")
def rng_next_op(state: Tensor) -> Tensor:
    """Advance the persistent {seed, offset} state by one step and return this step's pair (int64[2])."""
    out = torch.empty(2, dtype=torch.int64, device=state.device)
    with torch.cuda.device(state.device):
        _lib.check(_lib.lib().cgvp_rng_next(_ptr(state), _ptr(out), _stream()), "cgvp_rng_next")
    return out


@rng_next_op.register_fake
def _(state):
    return state.new_empty((2,))


def _rng(pair, p, site):
    return C.byref(Rng(pair.data_ptr(), float(p), int(site))) if p > 0 else None


def _rows(t):
    return int(t.numel() // t.shape[-1]), int(t.shape[-1])


@torch.library.custom_op("caster_gvp::dropout_add", mutates_args=(), device_types="cuda")
def dropout_add_op(a: Tensor, x: Tensor, pair: Tensor, site: int, p: float) -> Tensor:
    """x + dropout_p(a); an empty `x` means dropout_p(a) alone."""
    a = _f32(a, "a")
    x = _f32(x, "x") if x.numel() else None
    R, D = _rows(a)
    y = torch.empty_like(a)
    with torch.cuda.device(a.device):
        _lib.check(_lib.lib().cgvp_dropout_add(_ptr(a), _ptr(x), _rng(pair, p, site), R, D, _ptr(y), _stream()), "cgvp_dropout_add")
    return y


@dropout_add_op.register_fake
def _(a, x, pair, site, p):
    return torch.empty_like(a)


@torch.library.custom_op("caster_gvp::dropout_scale", mutates_args=(), device_types="cuda")
def dropout_scale_op(g: Tensor, pair: Tensor, site: int, p: float) -> Tensor:
    g = _f32(g, "grad_output")
    R, D = _rows(g)
    out = torch.empty_like(g)
    with torch.cuda.device(g.device):
        _lib.check(_lib.lib().cgvp_dropout_scale(_ptr(g), _rng(pair, p, site), R, D, _ptr(out), _stream()), "cgvp_dropout_scale")
    return out


@dropout_scale_op.register_fake
def _(g, pair, site, p):
    return torch.empty_like(g)


def _da_setup(ctx, inputs, output):
    a, x, pair, site, p = inputs
    ctx.save_for_backward(pair)
    ctx.site, ctx.p, ctx.has_x = site, p, bool(x.numel())


def _da_backward(ctx, g):
    (pair,) = ctx.saved_tensors
    ga = torch.ops.caster_gvp.dropout_scale(g.contiguous(), pair, ctx.site, ctx.p) if ctx.needs_input_grad[0] else None
    return ga, (g if (ctx.has_x and ctx.needs_input_grad[1]) else None), None, None, None


torch.library.register_autograd("caster_gvp::dropout_add", _da_backward, setup_context=_da_setup)


@torch.library.custom_op("caster_gvp::act_dropout", mutates_args=(), device_types="cuda")
def act_dropout_op(t: Tensor, pair: Tensor, site: int, p: float, slope: float) -> Tensor:
    """dropout_p(LeakyReLU_slope(t)); slope 0 = ReLU."""
    t = _f32(t, "t")
    R, D = _rows(t)
    y = torch.empty_like(t)
    with torch.cuda.device(t.device):
        _lib.check(_lib.lib().cgvp_act_dropout_fwd(_ptr(t), _rng(pair, p, site), float(slope), R, D, _ptr(y), _stream()),
                   "cgvp_act_dropout_fwd")
    return y


@act_dropout_op.register_fake
def _(t, pair, site, p, slope):
    return torch.empty_like(t)


@torch.library.custom_op("caster_gvp::act_dropout_backward", mutates_args=(), device_types="cuda")
def act_dropout_backward_op(g: Tensor, y: Tensor, pair: Tensor, site: int, p: float, slope: float) -> Tensor:
    g, y = _f32(g, "grad_output"), _f32(y, "y")
    R, D = _rows(g)
    gt = torch.empty_like(g)
    with torch.cuda.device(g.device):
        _lib.check(_lib.lib().cgvp_act_dropout_bwd(_ptr(g), _ptr(y), _rng(pair, p, site), float(slope), R, D, _ptr(gt), _stream()),
                   "cgvp_act_dropout_bwd")
    return gt


@act_dropout_backward_op.register_fake
def _(g, y, pair, site, p, slope):
    return torch.empty_like(g)


def _ad_setup(ctx, inputs, output):
    t, pair, site, p, slope = inputs
    ctx.save_for_backward(output, pair)
    ctx.site, ctx.p, ctx.slope = site, p, slope


def _ad_backward(ctx, g):
    y, pair = ctx.saved_tensors
    return torch.ops.caster_gvp.act_dropout_backward(g.contiguous(), y, pair, ctx.site, ctx.p, ctx.slope), None, None, None, None


torch.library.register_autograd("caster_gvp::act_dropout", _ad_backward, setup_context=_ad_setup)


class DropSites:
    """The fused dropout sites of ONE training forward of the head: one generator advance per step (`pair`), a fresh
    stream id per call site (call order is the same every step, so a site keeps its id).  `active` is False in eval
    mode, with p = 0, under torch.compile / autocast or off the GPU -- the callers then use the stock ops."""

    def __init__(self, rows, training, p):
        self.p = float(p)
        self.active = bool(training and p > 0 and rows.is_cuda and rows.dtype == torch.float32
                           and not torch.compiler.is_compiling() and not torch.is_autocast_enabled("cuda")
                           and torch.is_grad_enabled())
        self.site = 0
        self._br = _bridge() if self.active else None
        if not self.active:
            self.pair = None
        elif self._br is not None:
            self.pair = self._br.head_rng_next(_head_rng_state(rows.device))
        else:
            self.pair = torch.ops.caster_gvp.rng_next(_head_rng_state(rows.device))
        self._none = rows.new_empty(0) if self.active else None

    def _ok(self, t):
        return self.active and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] % 8 == 0 \
            and t.is_contiguous() and t.data_ptr() % 16 == 0

    def _next(self):
        self.site += 1
        return self.site

    def dropout_add(self, x, a, dropout):
        """x + dropout(a)."""
        if self._ok(a) and self._ok(x) and abs(dropout.p - self.p) < 1e-12:
            if self._br is not None:
                return self._br.head_dropout_add(a, x, self.pair, self._next(), self.p)
            return torch.ops.caster_gvp.dropout_add(a, x, self.pair, self._next(), self.p)
        return x + dropout(a)

    def act_dropout(self, t, activation, dropout):
        """dropout(activation(t)) for ReLU / LeakyReLU activations."""
        slope = _slope(activation)
        if slope is not None and self._ok(t) and abs(dropout.p - self.p) < 1e-12:
            if self._br is not None:
                return self._br.head_act_dropout(t, self.pair, self._next(), self.p, slope)
            return torch.ops.caster_gvp.act_dropout(t, self.pair, self._next(), self.p, slope)
        return dropout(activation(t))


def _slope(activation):
    if isinstance(activation, torch.nn.ReLU) or activation is torch.relu or activation is torch.nn.functional.relu:
        return 0.0
    if isinstance(activation, torch.nn.LeakyReLU):
        return float(activation.negative_slope)
    return None
