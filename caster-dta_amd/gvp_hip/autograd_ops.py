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
        h_last = torch.empty(N, ROW, dtype=torch.float32, device=dev)     # input of the head, saved for its backward
        with torch.cuda.device(dev):
            st = _stream()
            d, lay, P, I = C.byref(dims), C.byref(layout), _ptr(m["params"]), _ptr(image)
            _lib.check(L.cgvp_node_embed_fwd(d, lay, P, I, _ptr(x_s), _ptr(x_v), _ptr(nt), N, _ptr(hs[0]), st),
                       "cgvp_node_embed_fwd")
            for l in range(nc):
                last = l == nc - 1
                if ops.fuse_layer(N, E):
                    with ops._timed("conv_fwd"):
                        _lib.check(L.cgvp_conv_layer_fwd(d, lay, I, l, _ptr(hs[l]), _ptr(e_s), _ptr(e_v), _ptr(et),
                                                         _ptr(csr.rowptr), _ptr(csr.eperm), _ptr(csr.esrc),
                                                         _ptr(csr.edst), N, E, 1 if m["mean"] else 0,
                                                         _ptr(masks[l][0]), _ptr(masks[l][1]), 1 if last else 0,
                                                         _ptr(dhs[l]), _ptr(h_last if last else hs[l + 1]), _ptr(out),
                                                         st), "cgvp_conv_layer_fwd")
                    continue
                with ops._timed("conv_fwd"):
                    _lib.check(L.cgvp_conv_fwd(d, lay, P, I, l, _ptr(hs[l]), _ptr(e_s), _ptr(e_v), _ptr(et),
                                               _ptr(csr.rowptr), _ptr(csr.eperm), _ptr(csr.esrc), _ptr(csr.edst),
                                               N, E, 1 if m["mean"] else 0, _ptr(dhs[l]), st), "cgvp_conv_fwd")
                _lib.check(L.cgvp_node_update_fwd_train(d, lay, I, l, _ptr(hs[l]), _ptr(dhs[l]), _ptr(masks[l][0]),
                                                        _ptr(masks[l][1]), N, 1 if last else 0,
                                                        _ptr(h_last if last else hs[l + 1]), _ptr(out), st),
                           "cgvp_node_update_fwd_train")
        ctx.meta = m
        ctx.saved = (x_s, x_v, e_s, e_v, nt, et, hs, dhs, masks, h_last)
        ctx.n_params = len(params)
        return out

    @staticmethod
    def backward(ctx, g_out):
        L = _lib.lib()
        m = ctx.meta
        dims, layout, image, csr = m["dims"], m["layout"], m["image"], m["csr"]
        x_s, x_v, e_s, e_v, nt, et, hs, dhs, masks, h_last = ctx.saved
        N, E = int(x_s.shape[0]), int(e_s.shape[0])
        dev = x_s.device
        nc = m["num_convs"]
        g_out = _f32(g_out, "grad_output")
        f32 = dict(dtype=torch.float32, device=dev)
        gparams = torch.zeros(layout.total, **f32)
        # every stage writes its per-workgroup partial weight-gradient blocks into its own region of
        # one workspace; a single reduce launch at the end sums them all in a fixed order
        wsz = int(L.cgvp_bwd_workspace_floats(C.byref(dims), C.byref(layout)))
        nstage = 2 * nc + 1
        ws_all = torch.empty(nstage * wsz, **f32)
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
                g_src = torch.empty(N, ROW, **f32)       # zeroed by the node stage, filled by the conv stage's atomics
                _lib.check(L.cgvp_node_update_bwd(d, lay, I, l, _ptr(hs[l]), _ptr(dhs[l]), _ptr(masks[l][0]),
                                                  _ptr(masks[l][1]), _ptr(h_last if last else None),
                                                  _ptr(g_out if last else None), _ptr(ups[0]),
                                                  _ptr(ups[1]), _ptr(ups[2]), N, 1 if last else 0, _ptr(g_dh),
                                                  _ptr(g_h), _ptr(g_src), _ptr(gparams), _ptr(region()),
                                                  C.byref(segs, nseg * C.sizeof(_lib.Segment)), C.byref(cnt), st),
                           "cgvp_node_update_bwd")
                take()
                g_dst = torch.empty(N, ROW, **f32)
                with ops._timed("conv_bwd"):
                    _lib.check(L.cgvp_conv_bwd(d, lay, I, l, _ptr(hs[l]), _ptr(e_s), _ptr(e_v), _ptr(et),
                                               _ptr(csr.rowptr), _ptr(csr.eperm), _ptr(csr.esrc), _ptr(csr.edst), N, E,
                                               1 if m["mean"] else 0, _ptr(g_dh), _ptr(g_src), 1, _ptr(g_dst),
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
            _lib.check(L.cgvp_bwd_reduce(segs, nseg, _ptr(gparams), st), "cgvp_bwd_reduce")
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


_GINE_KEYS = ("eps", "w0", "b0", "w1", "b1", "we", "be")     # slab / state_dict order of one GINEConv


class _GineEncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, x, eattr, *params):
        m = meta
        widths, nl = m["widths"], len(m["widths"]) - 1
        x, eattr = _f32(x, "x"), _f32(eattr, "eattr")
        N = int(x.shape[0])
        ws = [dict(zip(_GINE_KEYS, params[7 * l:7 * l + 7])) for l in range(nl)]
        hs, masks = [x], []
        for l in range(nl):
            mask = None
            if m["dropout"] > 0 and l < nl - 1:
                keep = 1.0 - m["dropout"]
                mask = (torch.rand(N, widths[l + 1], device=x.device) < keep).to(torch.float32).div_(keep)
            masks.append(mask)
            first = l == 0
            hs.append(ops.gine_conv_forward(hs[l], m["ntypes"] if first else None, m["num_ntypes"] if first else 0,
                                            eattr, m["etypes"], m["num_etypes"], m["csr"], ws[l], widths[l],
                                            widths[l + 1], widths[l + 1], m["slope"], mask=mask))
        ctx.meta, ctx.saved = m, (hs, masks, eattr, ws)
        return hs[-1]

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        m = ctx.meta
        hs, masks, eattr, ws = ctx.saved
        widths, nl = m["widths"], len(m["widths"]) - 1
        csr = m["csr"]
        N, dev = int(hs[0].shape[0]), hs[0].device
        f32 = dict(dtype=torch.float32, device=dev)
        nt = _i64(m["ntypes"], "ntypes")
        et = _i64(m["etypes"], "etypes")
        g = _f32(g, "grad_output")
        wsp = torch.empty(int(L.cgvp_gine_bwd_workspace_floats()), **f32)
        grads = [None] * (7 * nl)
        edge_dim = int(eattr.shape[1])
        with torch.cuda.device(dev):
            for l in reversed(range(nl)):
                first = l == 0
                cin, cout = widths[l], widths[l + 1]
                w = {k: _f32(v, k) for k, v in ws[l].items()}
                sizes = [w[k].numel() for k in _GINE_KEYS]
                glayer = torch.zeros(sum(sizes), **f32)
                need_x = (not first) or ctx.needs_input_grad[1]
                g_x = torch.empty(N, cin - (m["num_ntypes"] if first else 0), **f32) if need_x else None
                gw = _lib.GineW(**{k: v.data_ptr() for k, v in w.items()})
                rc = L.cgvp_gine_conv_bwd(_ptr(hs[l]), _ptr(nt if first else None), m["num_ntypes"] if first else 0,
                                          _ptr(eattr), _ptr(et), m["num_etypes"], edge_dim, _ptr(csr.rowptr),
                                          _ptr(csr.eperm), _ptr(csr.esrc), _ptr(csr.edst), N, csr.num_edges, cin, cout, cout,
                                          C.byref(gw), float(m["slope"]), _ptr(masks[l]), _ptr(g), _ptr(g_x),
                                          _ptr(glayer), _ptr(wsp), int(m.get("bwd_workgroups", 0)), _stream())
                _lib.check(rc, "cgvp_gine_conv_bwd")
                off = 0
                for j, k in enumerate(_GINE_KEYS):
                    grads[7 * l + j] = glayer[off:off + sizes[j]].view(ws[l][k].shape)
                    off += sizes[j]
                g = g_x
        return (None, g if ctx.needs_input_grad[1] else None, None) + tuple(grads)


def gine_encoder(model, x, ntypes, eattr, etypes, csr, slope, train_dropout, bwd_workgroups=0):
    """HomoMoleculeGNN_GINE.forward with autograd (and inter-layer dropout when training)."""
    if eattr.requires_grad:
        raise NotImplementedError("gradients w.r.t. bond features are not produced by the backward kernels")
    params = []
    for conv in model.conv_list:
        kw = conv.kernel_weights()
        params += [kw[k] for k in _GINE_KEYS]
    meta = dict(widths=model._widths, ntypes=ntypes, etypes=etypes, num_ntypes=model.num_ntypes,
                num_etypes=model.num_etypes, csr=csr, slope=slope,
                dropout=float(model.dropout_rate) if train_dropout else 0.0, bwd_workgroups=bwd_workgroups)
    return _GineEncoderFn.apply(meta, x, eattr, *params)
