"""torch.autograd integration of the HIP kernels (training path).

`lba_encoder` / `gine_encoder` run the same forward kernels as inference while
saving each stage's INPUTS (node rows h_l, aggregated messages dh_l, dropout
masks); the backward launches the hand-written backward kernels, which
recompute their stage and emit data gradients plus an arena of weight gradients
whose views are handed back to autograd, one per nn.Parameter.  No PyTorch
arithmetic is involved besides drawing dropout masks.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, ops
from .ops import ROW, _f32, _i64, _ptr, _stream

MROW = 20   # dropout mask row: 16 scalar-channel + 4 vector-channel factors


def _dropout_mask(n, p, device):
    keep = 1.0 - p
    return (torch.rand(n, MROW, device=device) < keep).to(torch.float32).div_(keep)


class _LbaEncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, x_s, x_v, e_s, e_v, *params):
        L = _lib.lib()
        m = meta
        dims, layout, image, csr = m["dims"], m["layout"], m["image"], m["csr"]
        x_s, x_v, e_s, e_v = _f32(x_s, "x_s"), _f32(x_v, "x_v"), _f32(e_s, "eattr_s"), _f32(e_v, "eattr_v")
        N, E = int(x_s.shape[0]), int(e_s.shape[0])
        nt = _i64(m["ntypes"], "ntypes") if layout.nt_node > 0 else None
        et = _i64(m["etypes"], "etypes") if layout.nt_edge > 0 else None
        dev = x_s.device
        nc = m["num_convs"]
        hs = [torch.empty(N, ROW, dtype=torch.float32, device=dev) for _ in range(nc)]
        dhs = [torch.empty(N, ROW, dtype=torch.float32, device=dev) for _ in range(nc)]
        masks = [(None, None)] * nc
        if m["dropout"] > 0:
            masks = [(_dropout_mask(N, m["dropout"], dev), _dropout_mask(N, m["dropout"], dev)) for _ in range(nc)]
        out = torch.empty(N, dims.out_s, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _stream()
            d, lay, P, I = C.byref(dims), C.byref(layout), _ptr(m["params"]), _ptr(image)
            _lib.check(L.cgvp_node_embed_fwd(d, lay, P, I, _ptr(x_s), _ptr(x_v), _ptr(nt), N, _ptr(hs[0]), st),
                       "cgvp_node_embed_fwd")
            for l in range(nc):
                _lib.check(L.cgvp_conv_fwd(d, lay, P, I, l, _ptr(hs[l]), _ptr(e_s), _ptr(e_v), _ptr(et),
                                           _ptr(csr.rowptr), _ptr(csr.eperm), _ptr(csr.esrc), _ptr(csr.edst),
                                           N, E, 1 if m["mean"] else 0, _ptr(dhs[l]), st), "cgvp_conv_fwd")
                last = l == nc - 1
                _lib.check(L.cgvp_node_update_fwd_train(d, lay, I, l, _ptr(hs[l]), _ptr(dhs[l]), _ptr(masks[l][0]),
                                                        _ptr(masks[l][1]), N, 1 if last else 0,
                                                        _ptr(None if last else hs[l + 1]), _ptr(out), st),
                           "cgvp_node_update_fwd_train")
        ctx.meta = m
        ctx.saved = (x_s, x_v, e_s, e_v, nt, et, hs, dhs, masks)
        ctx.n_params = len(params)
        return out

    @staticmethod
    def backward(ctx, g_out):
        L = _lib.lib()
        m = ctx.meta
        dims, layout, image, csr = m["dims"], m["layout"], m["image"], m["csr"]
        x_s, x_v, e_s, e_v, nt, et, hs, dhs, masks = ctx.saved
        N, E = int(x_s.shape[0]), int(e_s.shape[0])
        dev = x_s.device
        nc = m["num_convs"]
        g_out = _f32(g_out, "grad_output")
        f32 = dict(dtype=torch.float32, device=dev)
        gparams = torch.zeros(layout.total, **f32)
        ws = torch.empty(int(L.cgvp_bwd_workspace_floats(C.byref(dims), C.byref(layout))), **f32)
        need_x = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        g_x_s = torch.empty(N, dims.node_in_s, **f32) if need_x else None
        g_x_v = torch.empty(N, dims.node_in_v, 3, **f32) if need_x else None
        with torch.cuda.device(dev):
            st = _stream()
            d, lay, I = C.byref(dims), C.byref(layout), _ptr(image)
            ups = (None, None, None)        # gradient w.r.t. the output of layer l (sum of up to 3 buffers)
            for l in reversed(range(nc)):
                last = l == nc - 1
                g_dh = torch.empty(N, ROW, **f32)
                g_h = torch.empty(N, ROW, **f32) if masks[l][0] is not None else None
                _lib.check(L.cgvp_node_update_bwd(d, lay, I, l, _ptr(hs[l]), _ptr(dhs[l]), _ptr(masks[l][0]),
                                                  _ptr(masks[l][1]), _ptr(g_out if last else None), _ptr(ups[0]),
                                                  _ptr(ups[1]), _ptr(ups[2]), N, 1 if last else 0, _ptr(g_dh),
                                                  _ptr(g_h), _ptr(gparams), _ptr(ws), st), "cgvp_node_update_bwd")
                g_src = torch.empty(N, ROW, **f32)
                g_dst = torch.empty(N, ROW, **f32)
                _lib.check(L.cgvp_conv_bwd(d, lay, I, l, _ptr(hs[l]), _ptr(e_s), _ptr(e_v), _ptr(et),
                                           _ptr(csr.rowptr), _ptr(csr.eperm), _ptr(csr.esrc), _ptr(csr.edst), N, E,
                                           1 if m["mean"] else 0, _ptr(g_dh), _ptr(g_src), _ptr(g_dst),
                                           _ptr(gparams), _ptr(ws), st), "cgvp_conv_bwd")
                ups = (g_h if g_h is not None else g_dh, g_src, g_dst)
            _lib.check(L.cgvp_node_embed_bwd(d, lay, I, _ptr(x_s), _ptr(x_v), _ptr(nt), N, _ptr(ups[0]), _ptr(ups[1]),
                                             _ptr(ups[2]), _ptr(g_x_s), _ptr(g_x_v), _ptr(gparams), _ptr(ws), st),
                       "cgvp_node_embed_bwd")
        grads = m["arena"].split(gparams)
        return (None, g_x_s, g_x_v, None, None) + tuple(grads)


def lba_encoder(model, params, layout, dims, x_s, x_v, ntypes, e_s, e_v, etypes, csr, train_dropout):
    """VectorProteinGNN_LBAModel.forward with autograd (and dropout when training)."""
    if ops.VARIANT != "mfma":
        raise NotImplementedError("training / gradients need the MFMA kernels (CGVP_VARIANT=mfma)")
    if e_s.requires_grad or e_v.requires_grad:
        raise NotImplementedError("gradients w.r.t. raw edge features are not produced by the backward kernels")
    meta = dict(dims=dims, layout=layout, image=model._fragment_image(params, layout, dims), csr=csr,
                params=params, arena=model._arena, ntypes=ntypes, etypes=etypes, num_convs=model.num_convs,
                mean=(model.aggr == "mean"), dropout=float(model.dropout_rate) if train_dropout else 0.0)
    return _LbaEncoderFn.apply(meta, x_s, x_v, e_s, e_v, *model._arena.params)


def gine_encoder(model, x, ntypes, eattr, etypes, csr, slope, train_dropout):
    raise NotImplementedError("GINE backward kernels are not wired up yet")
