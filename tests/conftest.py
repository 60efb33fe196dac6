import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "caster-dta_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def pretrained():
    """Full pretrained state dict (weights as data), keys without `_orig_mod.`."""
    return {k: torch.from_numpy(v) for k, v in load_npz("pretrained_state.npz").items()}


@pytest.fixture(scope="session")
def protein_params(pretrained):
    pfx = "protein_gnn.gnn_model."
    return {k[len(pfx):]: v for k, v in pretrained.items() if k.startswith(pfx)}


@pytest.fixture(scope="session")
def molecule_params(pretrained):
    pfx = "molecule_gnn.gnn_model."
    return {k[len(pfx):]: v for k, v in pretrained.items() if k.startswith(pfx)}


@pytest.fixture(scope="session")
def lba_small():
    return load_npz("lba_small.npz")


@pytest.fixture(scope="session")
def lba_sparse():
    return load_npz("lba_sparse.npz")


@pytest.fixture(scope="session")
def gvp_units():
    return load_npz("gvp_units.npz")


def rel_err(a, b):
    """max-abs error relative to the reference's max-abs (the parity metric of
    BASELINE.json: <= 1e-4 relative fp32)."""
    a = a.detach().double() if isinstance(a, torch.Tensor) else torch.as_tensor(a).double()
    b = b.detach().double() if isinstance(b, torch.Tensor) else torch.as_tensor(b).double()
    if b.numel() == 0:
        return 0.0
    return float((a.cpu() - b.cpu()).abs().max() / b.abs().max().clamp_min(1e-30))
