"""The ctypes stub of INTEGRATION.md section 2 (examples/caster_gvp_ctypes.py: torch + ctypes only, no gvp_hip import)
is the text in the document and actually runs: reference golden output through it."""
import importlib.util
import os

import pytest
import torch

from conftest import REPO, rel_err


def _stub_source():
    return open(os.path.join(REPO, "examples", "caster_gvp_ctypes.py")).read()


def test_integration_md_carries_the_tested_stub():
    md = open(os.path.join(REPO, "INTEGRATION.md")).read()
    assert _stub_source().strip() in md


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
