"""The golden recipe must keep running against the REFERENCE (not against this repo's own `models` package)
and must reproduce the committed fixtures bit for bit.  Needs /root/reference: runs in the build container,
skipped on the GPU box (which only reads the committed .npz files)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN

REF = "/root/reference"
FILES = ("lba_small.npz", "lba_sparse.npz", "gvp_units.npz", "pretrained_state.npz", "gvp_stacks.npz", "edge_feats.npz",
         "lba_amp_bf16.npz")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")), reason="reference tree not present")
def test_recipe_regenerates_committed_fixtures_bitwise(tmp_path):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}      # nothing of this repo on the path
    res = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py"), "--out-dir", str(tmp_path)],
                         cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "All keys matched" in res.stdout
    for name in FILES:
        new, old = np.load(tmp_path / name), np.load(os.path.join(GOLDEN, name))
        assert set(new.files) == set(old.files), name
        for k in old.files:
            assert new[k].dtype == old[k].dtype and new[k].shape == old[k].shape, (name, k)
            assert new[k].tobytes() == old[k].tobytes(), (name, k)


def test_recipe_never_imports_this_repo():
    """Static guard for the failure VERDICT r01 found: caster-dta_amd/ on sys.path makes `import models`
    resolve to the product (regular package beats the reference's namespace package)."""
    src = open(os.path.join(GOLDEN, "make_golden.py")).read()
    assert "sys.path.insert(0, os.path.join(REPO" not in src
    assert 'startswith(REF + "/")' in src


def test_sparse_golden_is_in_the_fused_regime(lba_sparse):
    g = lba_sparse
    assert g["edge_index"].shape[1] <= 4 * g["x_s"].shape[0]        # gvp_hip.ops.fuse_layer's condition
    assert len(g["ptr"]) == 4 and len({int(b - a) for a, b in zip(g["ptr"][:-1], g["ptr"][1:])}) == 3   # ragged
