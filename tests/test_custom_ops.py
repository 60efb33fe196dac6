"""The encoders as torch.library custom ops: the reference compiles its model
(train_model.py:422 torch.compile(model, dynamic=True)), runs it under autocast (:561) and scales the loss with a
GradScaler (:478, :570-587).  CPU part: schema + fake kernels propagate shapes without a GPU.  GPU part: opcheck,
compiled == eager, autocast + GradScaler step == fp32 step."""
import json
import os

import numpy as np
import pytest
import torch

import davis_synth as ds
from conftest import GOLDEN, rel_err
from gvp_hip import autograd_ops  # noqa: F401  (registers the ops)

DEV = "cuda:0"
CFG = [17, 3, 32, 1, 16, 4, 32, 1, 64, 20, 1, 2, 0]


def _to(d, dev=DEV):
    return {k: (tuple(t.to(dev) for t in v) if isinstance(v, tuple) else v.to(dev)) for k, v in d.items()}


def _protein(state, train=False):
    from models.protein_gnn import SelectableProteinModelWrapper
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["protein_gnn_kwargs"]
    for k in ("in_channels", "edge_dim", "hidden_channels", "edge_hidden_channels"):
        kw[k] = tuple(kw[k])
    m = SelectableProteinModelWrapper(**kw)
    m.load_state_dict({"gnn_model." + k: v for k, v in state.items()})
    return m.to(DEV).train(train)


def _molecule(state, train=False):
    from models.molecule_gnn import SelectableMoleculeModelWrapper
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["molecule_gnn_kwargs"]
    m = SelectableMoleculeModelWrapper(**kw)
    m.load_state_dict({"gnn_model." + k: v for k, v in state.items()})
    return m.to(DEV).train(train)


# ------------------------------------------------------------------------------------------------ CPU
def test_ops_are_registered_with_schemas():
    for name in ("lba_encoder", "gine_encoder", "gine_encoder_backward"):
        op = getattr(torch.ops.caster_gvp, name).default
        assert "Tensor[] params" in str(op._schema)
    # the protein backward needs no parameters: the fragment image of the weights the forward ran with is in `ws`
    assert "Tensor ws" in str(torch.ops.caster_gvp.lba_encoder_backward.default._schema)
    assert "Tensor[] csr" in str(torch.ops.caster_gvp.lba_encoder.default._schema)


def test_fake_kernels_propagate_shapes_without_a_gpu(protein_params, molecule_params):
    """What Dynamo / AOTAutograd run at trace time: no kernel, shapes and dtypes only."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from gvp_hip.arena import lba_param_keys
    with FakeTensorMode():
        f = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device="cuda")
        params = [f(*protein_params[k].shape) for k in lba_param_keys(2) if protein_params[k].numel()]
        N, E = 77, 250
        args = (params, f(N, 17), f(N, 3, 3), f(N, dt=torch.int64), f(E, 32), f(E, 1, 3), f(E, dt=torch.int64),
                f(2, E, dt=torch.int64), [], CFG)
        out, ws, masks = torch.ops.caster_gvp.lba_encoder(*args, 0.2, True)
        assert out.shape == (N, 64) and masks.numel() == 0                 # masks live in the kernels
        assert ws.dtype == torch.uint8 and ws.shape == (_c_lba_ws_bytes(N, E, True),)
        out, ws0, masks = torch.ops.caster_gvp.lba_encoder(*args, 0.0, False)
        assert out.shape == (N, 64) and ws0.numel() == 0 and masks.numel() == 0
        g, gxs, gxv, ges, gev = torch.ops.caster_gvp.lba_encoder_backward(f(N, 64), *args[1:9], ws, f(0), CFG, 0.2, True, True)
        assert g.shape == (15117,) and gxs.shape == (N, 17) and gxv.shape == (N, 3, 3)
        assert ges.shape == (E, 32) and gev.shape == (E, 1, 3)
        g, gxs, gxv, ges, gev = torch.ops.caster_gvp.lba_encoder_backward(f(N, 64), *args[1:9], ws, f(0), CFG, 0.2, False, False)
        assert gxs.numel() == gxv.numel() == ges.numel() == gev.numel() == 0
        keys = ("eps", "nn.lins.0.weight", "nn.lins.0.bias", "nn.lins.1.weight", "nn.lins.1.bias", "lin.weight", "lin.bias")
        mparams = [f(*molecule_params[f"conv_list.{l}.{k}"].shape) for l in range(2) for k in keys]
        Na, Ea = 40, 130
        margs = (mparams, f(Na, 41), f(Na, dt=torch.int64), f(Ea, 9), f(Ea, dt=torch.int64), f(2, Ea, dt=torch.int64), [],
                 [52, 16, 64], 11, 5, 0.01)
        out, mws, mk = torch.ops.caster_gvp.gine_encoder(*margs, 0.2, True)
        assert out.shape == (Na, 64) and mk[0].numel() == 0 and mws.shape == (_c_gine_ws_bytes(Na, Ea, [52, 16, 64], True, saved_only=True),)
        gflat, gx = torch.ops.caster_gvp.gine_encoder_backward(f(Na, 64), *margs[:7], mws, mk, [52, 16, 64],
                                                               11, 5, 0.01, 0.2, True, 0)
        assert gflat.shape == (7390,) and gx.shape == (Na, 41)


def _c_lba_ws_bytes(N, E, save, storage=0, nc=2):
    """cgvp_lba_fwd_workspace (host-only C function, no GPU)."""
    import ctypes as C
    from gvp_hip import _lib, ops
    dims = ops.make_dims(storage=storage)
    layout = ops.lba_layout(dims, 20, 1, nc)
    w = _lib.LbaFwdWs()
    assert _lib.lib().cgvp_lba_fwd_workspace(C.byref(dims), C.byref(layout), N, E, 1 if save else 0, C.byref(w)) == 0
    return int(w.total)


def _c_gine_ws_bytes(N, E, widths, save, saved_only=False):
    import ctypes as C
    from gvp_hip import _lib
    cfg = autograd_ops._gine_cfg(widths, 11, 5, 9, 0.01)
    w = _lib.GineFwdWs()
    assert _lib.lib().cgvp_gine_fwd_workspace(C.byref(cfg), N, E, 1 if save else 0, C.byref(w)) == 0
    return int(w.saved if saved_only else w.total)


def test_python_workspace_formulas_match_the_library():
    """The fake kernels size `ws` with Python mirrors of cgvp_*_fwd_workspace: they must agree for every N, E, depth,
    storage type and mode, or compiled graphs would allocate the wrong workspace."""
    from gvp_hip import ops
    for N, E in ((0, 0), (1, 0), (5, 3), (77, 250), (300, 898), (19200, 57484), (1001, 20020)):
        for nc in (1, 2, 4):
            for storage, es in ((0, 4), (1, 2)):
                for save in (True, False):
                    dims = ops.make_dims(storage=storage)
                    sdt = torch.bfloat16 if storage else torch.float32
                    img = autograd_ops._dims_layout([17, 3, 32, 1, 16, 4, 32, 1, 64, 20, 1, nc, 0], sdt)[2]
                    assert autograd_ops.lba_fwd_ws_bytes(N, E, nc, img, es, save) == _c_lba_ws_bytes(N, E, save, storage, nc)
        for widths in ([52, 64], [52, 16, 64], [52, 16, 16, 16, 64]):
            for save in (True, False):
                assert autograd_ops.gine_fwd_ws_bytes(N, E, widths, save) == _c_gine_ws_bytes(N, E, widths, save)
                assert autograd_ops.gine_fwd_ws_bytes(N, E, widths, save, True) == _c_gine_ws_bytes(N, E, widths, save, True)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_opcheck(protein_params, molecule_params):
    """torch.library.opcheck: schema, fake-vs-real agreement, autograd registration, AOT dispatch (dynamic)."""
    prot, mol = _protein(protein_params), _molecule(molecule_params)
    p, m = ds.pair_batch(2, 5, lengths=[30, 41])
    pd, md = _to(ds.to_torch(p)), _to(ds.to_torch(m))
    params = prot.gnn_model._arena.params
    xs = pd["x"][0].clone().requires_grad_()
    args = (params, xs, pd["x"][1], pd["ntypes"], pd["eattr"][0], pd["eattr"][1], pd["etypes"], pd["edge_index"], [],
            CFG, 0.0, True)
    torch.library.opcheck(torch.ops.caster_gvp.lba_encoder.default, args)
    mparams = [mol.gnn_model.conv_list[l].kernel_weights()[k] for l in range(2) for k in autograd_ops._GINE_KEYS]
    margs = (mparams, md["x"].clone().requires_grad_(), md["ntypes"], md["eattr"], md["etypes"], md["edge_index"], [],
             [52, 16, 64], 11, 5, 0.01, 0.0, True)
    torch.library.opcheck(torch.ops.caster_gvp.gine_encoder.default, margs)


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["aot_eager", "inductor"])
def test_compiled_equals_eager_bitwise(protein_params, molecule_params, backend):
    """torch.compile(model, dynamic=True) (train_model.py:422) over the encoders: same kernels, same bits --
    forward and every gradient -- and a second batch of another size runs without retracing errors."""
    prot, mol = _protein(protein_params), _molecule(molecule_params)
    cprot = torch.compile(prot, dynamic=True, backend=backend)
    cmol = torch.compile(mol, dynamic=True, backend=backend)
    for seed, lengths in ((3, [40, 66, 35]), (4, [120, 31])):
        p, m = ds.pair_batch(len(lengths), seed, lengths=lengths)
        pd, md = _to(ds.to_torch(p)), _to(ds.to_torch(m))
        r = torch.randn(p.num_nodes, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
        ra = torch.randn(m.num_nodes, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
        got = {}
        for tag, fp, fm in (("eager", prot, mol), ("compiled", cprot, cmol)):
            prot.zero_grad(), mol.zero_grad()
            res, atm = fp(**pd), fm(**md)
            ((res * r).sum() + (atm * ra).sum()).backward()
            got[tag] = (res.detach().clone(), atm.detach().clone(),
                        [q.grad.clone() for q in list(prot.parameters()) + list(mol.parameters()) if q.numel()])
        assert torch.equal(got["eager"][0], got["compiled"][0]) and torch.equal(got["eager"][1], got["compiled"][1])
        # weight gradients below the last conv layer pass through float atomics (d h[src]): equal to rounding
        # (floor tied to the global gradient scale: analytically-zero gradients -- gvp_edge's gate weights, ~1e-8 -- are pure
        # rounding noise that the atomics reorder)
        scale = max(float(a.abs().max()) for a in got["eager"][2])
        for a, b in zip(got["eager"][2], got["compiled"][2]):
            assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-7 * scale
    with torch.no_grad():                                    # inference launch sequence under compile
        assert torch.equal(cprot(**pd), prot(**pd))


@pytest.mark.gpu
def test_compiled_joint_model_train_step(pretrained):
    """The reference's training step with the model compiled: JointGNN (encoders = custom ops, head = stock torch)
    under torch.compile(dynamic=True), autocast and a GradScaler (train_model.py:422, :561, :570-587)."""
    from models.joint_gnn import JointGNN
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
    model = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"],
                     **kw["joint_gnn_kwargs"])
    model.load_state_dict(pretrained, strict=True)
    model = model.to(DEV).train()
    cmodel = torch.compile(model, dynamic=True)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    # a modest initial scale: the default 65536 makes the fp16 head overflow on some steps, which GradScaler handles by
    # skipping the step (by design) -- but then that step's gradients are legitimately non-finite
    scaler = torch.amp.GradScaler("cuda", init_scale=256.0)
    losses = []
    for step in range(6):
        p, m = ds.pair_batch(4, 30 + step % 2, lengths=[40 + 3 * step, 60, 33, 51])
        pd, md = _to(ds.to_torch(p)), _to(ds.to_torch(m))
        target = torch.linspace(-1, 1, 4, device=DEV).unsqueeze(-1)
        opt.zero_grad()
        with torch.autocast("cuda", enabled=True):
            pred, _ = cmodel(pd, md)
            loss = torch.nn.functional.mse_loss(pred.float(), target)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        scaler.step(opt)
        scaler.update()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses))
    missing = [n for n, q in model.named_parameters() if q.numel() and q.grad is None]
    assert not missing, missing
    assert all(torch.isfinite(q.grad).all() for q in model.parameters() if q.numel())


@pytest.mark.gpu
def test_autocast_and_gradscaler_equal_the_fp32_step(protein_params, molecule_params):
    """Under autocast the ops cast their float inputs to fp32 and compute exactly what they compute outside it;
    a GradScaler's power-of-two loss scale is undone exactly by unscale_."""
    prot, mol = _protein(protein_params), _molecule(molecule_params)
    p, m = ds.pair_batch(3, 8, lengths=[50, 44, 61])
    pd, md = _to(ds.to_torch(p)), _to(ds.to_torch(m))
    r = torch.randn(p.num_nodes, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)) * 1e-2
    ra = torch.randn(m.num_nodes, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2)) * 1e-2
    params = [q for q in list(prot.parameters()) + list(mol.parameters()) if q.numel()]

    def step(amp):
        for q in params:
            q.grad = None
        opt = torch.optim.SGD(params, lr=0.0)
        scaler = torch.amp.GradScaler("cuda", enabled=amp, init_scale=1024.0)
        with torch.autocast("cuda", enabled=amp):
            # a half-precision input (what an upstream autocast op would hand over) is cast back up by the op
            xin = (pd["x"][0].half().float(), pd["x"][1])
            res, atm = prot(**dict(pd, x=(xin[0].half(), xin[1]) if amp else xin)), mol(**md)
            assert res.dtype == torch.float32 and atm.dtype == torch.float32
            loss = (res * r).sum() + (atm * ra).sum()
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        return res.detach().clone(), atm.detach().clone(), [q.grad.clone() for q in params]

    a, b = step(False), step(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    scale = max(float(ga.abs().max()) for ga in a[2])
    for ga, gb_ in zip(a[2], b[2]):
        assert torch.isfinite(gb_).all()
        assert float((ga - gb_).abs().max()) <= 1e-5 * float(ga.abs().max()) + 1e-7 * scale
