"""The drop-in boundary on CPU: constructor kwargs, parameter names / shapes /
counts pinned by the reference's pretrained checkpoint, the C ABI's exported
symbols, and that the encoders refuse to run without the GPU library path."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import davis_synth as ds
from conftest import GOLDEN, REPO, rel_err
from gvp_hip import _lib, arena

T = torch.from_numpy


@pytest.fixture(scope="module")
def kwargs():
    return json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))


@pytest.fixture(scope="module")
def joint(kwargs, pretrained):
    from models.joint_gnn import JointGNN
    model = JointGNN(protein_gnn_kwargs=kwargs["protein_gnn_kwargs"],
                     molecule_gnn_kwargs=kwargs["molecule_gnn_kwargs"], **kwargs["joint_gnn_kwargs"])
    model.load_state_dict(pretrained, strict=True)        # inference_utils.py:67 contract
    return model.eval()


def test_checkpoint_contract(joint, pretrained):
    sd = joint.state_dict()
    assert set(sd) == set(pretrained)
    for k, v in pretrained.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    count = lambda m: sum(p.numel() for p in m.parameters())
    assert count(joint) == 764396                       # model_summary.txt:4
    assert count(joint.protein_gnn) == 15117            # model_summary.txt:7
    assert count(joint.molecule_gnn) == 7390            # model_summary.txt:17


def test_fused_parameter_mode_keeps_the_checkpoint_contract(kwargs, pretrained):
    """`JointGNN.fuse_encoder_parameters()` (opt-in, eager host time): the protein encoder owns ONE trainable leaf, its
    arena; `state_dict()` still has the reference's keys / shapes / values, `load_state_dict(strict=True)` still takes a
    reference checkpoint, per-module attributes stay readable views that follow an in-place update of the arena."""
    from models.joint_gnn import JointGNN
    make = lambda: JointGNN(protein_gnn_kwargs=kwargs["protein_gnn_kwargs"],
                            molecule_gnn_kwargs=kwargs["molecule_gnn_kwargs"], **kwargs["joint_gnn_kwargs"])
    model = make()
    model.load_state_dict(pretrained, strict=True)
    model.fuse_encoder_parameters()
    assert model.fuse_encoder_parameters() is model                       # idempotent
    gm = model.protein_gnn.gnn_model
    names = [k for k, p in model.protein_gnn.named_parameters() if p.numel()]
    assert names == ["gnn_model.arena"] and gm.arena.numel() == 15117
    dnames = [k for k, p in model.molecule_gnn.named_parameters() if p.numel()]
    assert dnames == ["gnn_model.arena"] and model.molecule_gnn.gnn_model.arena.numel() == 7390
    assert sum(p.numel() for p in model.parameters()) == 764396
    sd = model.state_dict()
    assert set(sd) == set(pretrained)
    for k, v in pretrained.items():
        assert tuple(sd[k].shape) == tuple(v.shape) and torch.equal(sd[k], v), k
    # a reference-format checkpoint loads into the fused model, strictly
    fresh = make().fuse_encoder_parameters()
    assert not torch.equal(fresh.protein_gnn.gnn_model.arena, gm.arena)
    fresh.load_state_dict(pretrained, strict=True)
    assert torch.equal(fresh.protein_gnn.gnn_model.arena, gm.arena)
    w = pretrained["protein_gnn.gnn_model.conv_list.1.conv.message_func.0.ws.weight"]
    assert torch.equal(fresh.protein_gnn.gnn_model.conv_list[1].conv.message_func[0].ws.weight, w)
    # ... and a fused model's checkpoint loads into an unfused one
    plain = make()
    plain.load_state_dict(sd, strict=True)
    # strict loading still reports what a checkpoint lacks / has too much of
    broken = {k: v for k, v in pretrained.items() if not k.endswith("gvp_to_scalar.ws.bias")}
    with pytest.raises(RuntimeError, match="gvp_to_scalar.ws.bias"):
        make().fuse_encoder_parameters().load_state_dict(broken, strict=True)
    # in-place update of the arena (what an optimizer does) is what the per-module views show
    with torch.no_grad():
        gm.arena.mul_(2.0)
        model.molecule_gnn.gnn_model.arena.add_(1.0)
    assert torch.equal(gm.conv_list[1].conv.message_func[0].ws.weight, 2.0 * w)
    assert torch.equal(model.molecule_gnn.gnn_model.conv_list[1].lin.bias,
                       pretrained["molecule_gnn.gnn_model.conv_list.1.lin.bias"] + 1.0)
    assert float(model.molecule_gnn.gnn_model.conv_list[0].eps) == float(pretrained["molecule_gnn.gnn_model.conv_list.0.eps"]) + 1.0
    # .double()/.float() round trip re-seats the views on the new storage
    model.double().float()
    assert gm.conv_list[0].norm[0].scalar_norm.weight.data_ptr() >= gm.arena.data_ptr()
    assert torch.equal(gm.conv_list[1].conv.message_func[0].ws.weight, 2.0 * w)


def test_wrapper_passthrough_and_errors(joint, kwargs):
    from models.protein_gnn import SelectableProteinModelWrapper
    from models.molecule_gnn import SelectableMoleculeModelWrapper
    assert joint.protein_gnn.out_channels == (64, 0) and joint.molecule_gnn.out_channels == 64
    assert joint.protein_gnn.num_ntypes == 20 and joint.molecule_gnn.num_etypes == 5
    pk = dict(kwargs["protein_gnn_kwargs"])
    with pytest.raises(ValueError):
        SelectableProteinModelWrapper(**dict(pk, in_channels=17, edge_dim=(32, 1)))
    with pytest.raises(ValueError):
        SelectableProteinModelWrapper(**dict(pk, in_channels=17, edge_dim=32))
    with pytest.raises(NotImplementedError):            # the non-GVP protein encoders stay out of scope
        SelectableProteinModelWrapper(**dict(pk, in_channels=17, edge_dim=32, base_conv="gatv2"))
    with pytest.raises(NotImplementedError):
        SelectableMoleculeModelWrapper(**dict(kwargs["molecule_gnn_kwargs"], base_conv="gatv2"))
    with pytest.raises(ValueError):
        from models.model_utils import _select_activation
        _select_activation("nope")


def test_arena_views_track_parameters(joint, protein_params):
    """Parameters become views of one flat buffer in state_dict order; optimiser-
    style in-place updates and load_state_dict land in the arena; re-materialised
    parameters (.double().float()) are detected and re-flattened."""
    enc = joint.protein_gnn.gnn_model
    named = dict(enc.named_parameters())
    ar = arena.ParamArena([named[k] for k in arena.lba_param_keys(2)])
    flat = ar.buffer()
    assert flat.numel() == 15117
    assert torch.equal(flat, arena.flatten_state(protein_params, 2))
    with torch.no_grad():
        named["conv_list.1.ff_func.0.ws.bias"].add_(1.0)
    assert ar.buffer() is flat
    off = ar.offsets[arena.lba_param_keys(2).index("conv_list.1.ff_func.0.ws.bias")]
    assert torch.allclose(flat[off:off + 64], protein_params["conv_list.1.ff_func.0.ws.bias"] + 1.0)
    enc.load_state_dict({k: v for k, v in protein_params.items()}, strict=False)
    assert torch.equal(ar.buffer(), arena.flatten_state(protein_params, 2))
    enc.double().float()
    named = dict(enc.named_parameters())
    flat2 = ar.buffer()
    assert flat2 is not flat and torch.equal(flat2, arena.flatten_state(protein_params, 2))
    grads = ar.split(torch.arange(15117.0))
    assert grads[1].shape == named["gvp_node.0.ws.weight"].shape and float(grads[1][0, 0]) == 12.0


def test_no_cpu_path(joint):
    p, m = ds.pair_batch(2, 0, lengths=[30, 20])
    with pytest.raises(RuntimeError, match="MI355X only"):
        joint(ds.to_torch(p), ds.to_torch(m))


def test_c_abi_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "caster_gvp.h")).read()
    declared = set(re.findall(r"\b(cgvp_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.exported_symbols())
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    assert _lib.lib().cgvp_abi_version() == _lib.ABI_VERSION


def test_standalone_module_compositions(gvp_units):
    """gvp_layers modules used on their own (not the fused path) follow the reference."""
    import models.gvp_layers as gvp
    u = gvp_units
    cases = {"msg0": ((64, 9), (16, 4), (F.relu, None), True), "ff1": ((64, 8), (16, 4), (None, None), True),
             "nogate_sigmoid": ((16, 4), (16, 4), (F.relu, torch.sigmoid), False),
             "scalar_only_in": ((12, 0), (8, 2), (F.relu, None), False),
             "to_scalar": ((16, 4), (64, 0), (F.relu, None), True)}
    for name, (din, dout, acts, gate) in cases.items():
        mod = gvp.GVP(din, dout, activations=acts, vector_gate=gate).eval()
        mod.load_state_dict({k[len(f"gvp_{name}_w_"):]: T(v) for k, v in u.items() if k.startswith(f"gvp_{name}_w_")})
        s, v = T(u[f"gvp_{name}_in_s"]), T(u[f"gvp_{name}_in_v"])
        out = mod((s, v) if din[1] else s)
        os_, ov = out if isinstance(out, tuple) else (out, None)
        assert rel_err(os_, u[f"gvp_{name}_out_s"]) < 1e-5, name
        if dout[1]:
            assert rel_err(ov, u[f"gvp_{name}_out_v"]) < 1e-5, name
    for aggr in ("sum", "mean"):
        key = f"convlayer_{aggr}_gate"
        layer = gvp.GVPConvLayer((16, 4), (32, 1), drop_rate=0.2, activations=(F.relu, None), vector_gate=True,
                                 aggr=aggr).eval()
        layer.load_state_dict({k[len(key + "_w_"):]: T(v) for k, v in u.items() if k.startswith(key + "_w_")})
        o = layer((T(u[key + "_in_s"]), T(u[key + "_in_v"])), T(u[key + "_edge_index"]),
                  (T(u[key + "_e_s"]), T(u[key + "_e_v"])))
        assert rel_err(o[0], u[key + "_out_s"]) < 1e-5 and rel_err(o[1], u[key + "_out_v"]) < 1e-5
    with pytest.raises(ValueError):
        gvp.GVPConvLayer((16, 4), (32, 1), autoregressive=True, aggr="mean")
    drop = gvp.Dropout(0.2).train()
    s, v = drop((torch.ones(4000, 16), torch.ones(4000, 4, 3)))
    assert set(np.unique(v.numpy()).round(4)) <= {0.0, 1.25}
    assert torch.equal(v[..., 0], v[..., 1]) and abs(float((v == 0).float().mean()) - 0.2) < 0.03


def test_to_dense_batch():
    from models.joint_gnn import to_dense_batch
    x = torch.arange(12.0).view(6, 2)
    dense, mask = to_dense_batch(x, torch.tensor([0, 0, 0, 1, 2, 2]))
    assert dense.shape == (3, 3, 2) and mask.tolist() == [[1, 1, 1], [1, 0, 0], [1, 1, 0]]
    assert torch.equal(dense[2, :2], x[4:6]) and float(dense[1, 1:].abs().sum()) == 0
