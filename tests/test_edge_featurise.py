"""On-device protein edge featurisation (csrc/feat_kernels.hip).

PINNED by tests/golden/edge_feats.npz: outputs of the reference's own `compute_residue_edge_features`
(utils/create_protein_features.py:201) + `construct_graph` (utils/create_graphs.py:6), imported unmodified by
tests/golden/make_golden.py -- self loops, coincident C-alphas, neighbours more than 1,000 apart in sequence, a kNN
graph.  The NumPy restatement below (float64, :225-273 + calc_pos_encoding :368-386) is itself checked against that
fixture on the CPU (test_numpy_restatement_matches_reference_fixture) and then serves at sizes the fixture does not hold.
Stated tolerance: 2e-6 absolute on the [-1, 1]-valued features (fp32 distance / exp / sincos against float64; the
positional-encoding argument is formed and range-reduced in fp64 on the device, so it holds for |j - i| in the
thousands as well); the unit directions to 2e-6, exact zeros on self loops and coincident atoms."""
import numpy as np
import pytest
import torch

import davis_synth as ds
from conftest import load_npz

DEV = "cuda:0"
GOLD_CASES = ("radius4", "coincident", "box_far_in_sequence", "knn12")


def numpy_edge_features(ca, seq, ei):
    i, j = ei[0], ei[1]
    diff = ca[i].astype(np.float64) - ca[j].astype(np.float64)            # row (source) minus column (destination), :244
    d = np.linalg.norm(diff, axis=-1)
    mu = np.linspace(0.0, 20.0, 16)                                       # :233-237
    rbf = np.exp(-np.square((d[:, None] - mu[None, :]) / (20.0 / 16)))
    freqs = np.exp(2 * np.arange(8) * -(np.log(10000.0) / 8))             # :379
    ang = (seq[j] - seq[i]).astype(np.float64)[:, None] * freqs           # destination index - source index, :254
    pos = np.concatenate([np.cos(ang), np.sin(ang)], axis=-1)
    with np.errstate(invalid="ignore", divide="ignore"):
        unit = np.where(d[:, None] > 0, diff / d[:, None], 0.0)           # normalize_vecs :360-365
    return np.concatenate([rbf, pos], -1), unit[:, None, :]


def test_numpy_restatement_matches_reference_fixture():
    """CPU: the restatement above == the reference's featuriser + COO construction on every fixture case."""
    g = load_npz("edge_feats.npz")
    for case in GOLD_CASES + ("synth64",):
        ca, ei = g[f"{case}_ca"], g[f"{case}_edge_index"]
        s, v = numpy_edge_features(ca, np.arange(ca.shape[0]), ei)
        assert float(np.abs(s - g[f"{case}_e_s"]).max()) < 1e-6, case          # the fixture is float32
        assert float(np.abs(v - g[f"{case}_e_v"]).max()) < 1e-6, case
        assert (ei[0, 1:] >= ei[0, :-1]).all()                                 # row-major COO: sources non-decreasing


def test_davis_synth_edges_are_the_reference_featuriser():
    """CPU: the synthetic generator every parity input comes from stores, for its own C-alpha trace, exactly the edge
    list and edge features the reference's featuriser produces (same threshold rule, ordering, self loops)."""
    g = load_npz("edge_feats.npz")
    mine = ds.protein_graph(50, np.random.default_rng(9), 5.0, "dist")
    assert np.array_equal(ds.ca_trace(50, np.random.default_rng(9)), g["synth64_ca"])
    assert np.array_equal(mine["edge_index"], g["synth64_edge_index"])
    assert float(np.abs(mine["e_s"] - g["synth64_e_s"]).max()) < 1e-6
    assert float(np.abs(mine["e_v"] - g["synth64_e_v"]).max()) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("case", GOLD_CASES)
def test_edge_features_match_reference_fixture(case):
    """GPU kernel vs the reference's own outputs."""
    from gvp_hip import ops
    g = load_npz("edge_feats.npz")
    ca, ei = g[f"{case}_ca"], g[f"{case}_edge_index"]
    assert ca.dtype == np.float32
    e_s, e_v = ops.edge_features(torch.from_numpy(ca).to(DEV), torch.arange(ca.shape[0], device=DEV),
                                 torch.from_numpy(ei).to(DEV))
    assert float(np.abs(e_s.cpu().numpy() - g[f"{case}_e_s"]).max()) < 2e-6
    assert float(np.abs(e_v.cpu().numpy() - g[f"{case}_e_v"]).max()) < 2e-6
    zero = np.abs(g[f"{case}_e_v"]).sum(axis=(1, 2)) == 0                    # self loops, coincident atoms
    assert zero.any() and float(e_v.cpu().numpy()[zero].__abs__().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["radius_batch", "knn_long", "far_apart_in_sequence"])
def test_edge_features_match_numpy(case):
    from gvp_hip import ops
    rng = np.random.default_rng(4)
    if case == "radius_batch":
        lens, thresh, kind = [40, 77, 33], 6.0, "dist"
    elif case == "knn_long":
        lens, thresh, kind = [600], 20, "num"
    else:
        lens, thresh, kind = [120], 4.0, "dist"
    cas, seqs, eis, off = [], [], [], 0
    for L in lens:
        ca = ds.ca_trace(L, rng).astype(np.float32)
        dist = np.linalg.norm(ca[:, None] - ca[None], axis=-1)
        if kind == "dist":
            keep = dist <= thresh
        else:
            keep = np.zeros_like(dist, dtype=bool)
            keep[np.arange(L)[:, None], np.argsort(dist, axis=-1, kind="stable")[:, :thresh]] = True
        src, dst = np.nonzero(keep)
        cas.append(ca); eis.append(np.stack([src, dst]) + off); off += L
        seqs.append(np.arange(L) if case != "far_apart_in_sequence" else np.arange(L) * 37)   # |j - i| up to ~4400
    ca, seq, ei = np.concatenate(cas), np.concatenate(seqs).astype(np.int64), np.concatenate(eis, 1).astype(np.int64)
    want_s, want_v = numpy_edge_features(ca, seq, ei)
    e_s, e_v = ops.edge_features(torch.from_numpy(ca).to(DEV), torch.from_numpy(seq).to(DEV), torch.from_numpy(ei).to(DEV))
    assert e_s.shape == (ei.shape[1], 32) and e_v.shape == (ei.shape[1], 1, 3)
    assert float(np.abs(e_s.cpu().numpy() - want_s).max()) < 2e-6
    assert float(np.abs(e_v.cpu().numpy() - want_v).max()) < 2e-6
    loops = ei[0] == ei[1]
    assert loops.any() and float(e_v.cpu().numpy()[loops].__abs__().max()) == 0.0
    # and it equals what the synthetic generator (the featuriser's shapes / ordering) stores for the same graph
    if case == "radius_batch":
        g = ds.protein_graph(50, np.random.default_rng(9), 5.0, "dist")
        # regenerate the same trace to feed the kernel
        ca2 = ds.ca_trace(50, np.random.default_rng(9)).astype(np.float32)
        s2, v2 = ops.edge_features(torch.from_numpy(ca2).to(DEV), torch.arange(50, device=DEV),
                                   torch.from_numpy(g["edge_index"]).to(DEV))
        assert float(np.abs(s2.cpu().numpy() - g["e_s"]).max()) < 5e-6 and float(np.abs(v2.cpu().numpy() - g["e_v"]).max()) < 5e-6


@pytest.mark.gpu
def test_featurised_edges_feed_the_encoder(protein_params):
    """coordinates -> device featuriser -> encoder == stored features -> encoder."""
    import json, os
    from conftest import GOLDEN, rel_err
    from gvp_hip import ops
    from models.protein_gnn import SelectableProteinModelWrapper
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["protein_gnn_kwargs"]
    for k in ("in_channels", "edge_dim", "hidden_channels", "edge_hidden_channels"):
        kw[k] = tuple(kw[k])
    model = SelectableProteinModelWrapper(**kw)
    model.load_state_dict({"gnn_model." + k: v for k, v in protein_params.items()})
    model = model.to(DEV).eval()
    g = ds.protein_graph(90, np.random.default_rng(2), 4.0, "dist")
    ca = ds.ca_trace(90, np.random.default_rng(2)).astype(np.float32)
    d = ds.to_torch(ds.collate([g]))
    dd = {k: (tuple(t.to(DEV) for t in v) if isinstance(v, tuple) else v.to(DEV)) for k, v in d.items()}
    e_s, e_v = ops.edge_features(torch.from_numpy(ca).to(DEV), torch.arange(90, device=DEV), dd["edge_index"])
    with torch.no_grad():
        a = model(dd["x"], dd["edge_index"], dd["ntypes"], dd["etypes"], eattr=dd["eattr"])
        b = model(dd["x"], dd["edge_index"], dd["ntypes"], dd["etypes"], eattr=(e_s, e_v))
    assert rel_err(b, a) < 2e-5
