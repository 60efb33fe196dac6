"""The ctypes stub of INTEGRATION.md section 2 (examples/caster_gvp_ctypes.py: torch + ctypes only, no gvp_hip import)
is the text in the document and actually runs: reference golden output through it."""
import importlib.util
import os

import pytest
import torch

from conftest import REPO, rel_err


def _stub_source(name="caster_gvp_ctypes.py"):
    return open(os.path.join(REPO, "examples", name)).read()


def test_integration_md_carries_the_tested_stub():
    md = open(os.path.join(REPO, "INTEGRATION.md")).read()
    assert _stub_source().strip() in md
    assert _stub_source("caster_gvp_pass_ctypes.py").strip() in md


@pytest.mark.gpu
def test_stub_reproduces_the_reference_output(lba_small, protein_params):
    spec = importlib.util.spec_from_file_location("caster_gvp_ctypes", os.path.join(REPO, "examples", "caster_gvp_ctypes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    enc = mod.ProteinEncoder(os.path.join(REPO, "caster-dta_amd", "lib", "libcaster_gvp.so"), protein_params)
    g = lba_small
    T = lambda a: torch.from_numpy(a).cuda()
    out = enc.forward(T(g["x_s"]), T(g["x_v"]), T(g["ntypes"]), T(g["edge_index"]), T(g["e_s"]), T(g["e_v"]), T(g["etypes"]))
    torch.cuda.synchronize()
    assert rel_err(out, g["out"]) < 2e-5


@pytest.mark.gpu
def test_whole_pass_stub_reproduces_reference_output_and_gradients(lba_sparse, protein_params):
    """The production entry points bound with ctypes alone (examples/caster_gvp_pass_ctypes.py): reference output and the
    reference's autograd gradients of every weight and of the node inputs, from ONE forward and ONE backward call."""
    from gvp_hip.arena import lba_param_keys
    spec = importlib.util.spec_from_file_location("caster_gvp_pass_ctypes", os.path.join(REPO, "examples", "caster_gvp_pass_ctypes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    enc = mod.ProteinEncoder(os.path.join(REPO, "caster-dta_amd", "lib", "libcaster_gvp.so"), protein_params)
    g = lba_sparse
    T = lambda a: torch.from_numpy(a).cuda()
    ins = (T(g["x_s"]), T(g["x_v"]), T(g["ntypes"]), T(g["edge_index"]), T(g["e_s"]), T(g["e_v"]), T(g["etypes"]))
    out, ws = enc.forward(*ins)
    gparams, g_xs, g_xv = enc.backward(T(g["r"]), ws, *ins)
    torch.cuda.synchronize()
    assert rel_err(out, g["out"]) < 2e-5
    assert rel_err(g_xs, g["gin_x_s"]) < 2e-4 and rel_err(g_xv, g["gin_x_v"]) < 2e-4
    want = torch.cat([torch.from_numpy(g["g_" + k]).reshape(-1) for k in lba_param_keys(2) if protein_params[k].numel()])
    scale = float(want.abs().max())
    assert float((gparams.cpu() - want).abs().max()) <= 2e-4 * scale
