"""GPU parity: HIP kernels called through the C ABI (gvp_hip.ops) against the
reference's golden vectors and against the CPU oracle on seeded Davis-shaped
inputs.  Bar: <= 1e-4 relative fp32 (BASELINE.json), tested at 2e-5."""
import numpy as np
import pytest
import torch

import davis_synth as ds
from conftest import rel_err
from gvp_hip import arena, ops
from oracle import gvp_oracle as O

pytestmark = pytest.mark.gpu
TOL = 2e-5
DEV = "cuda:0"


@pytest.fixture(autouse=True, params=["mfma", "simt"])
def variant(request):
    """Every test runs on both kernel families: the MFMA tile kernels (production)
    and the independent one-item-per-lane kernels."""
    old = ops.VARIANT
    ops.VARIANT = request.param
    yield request.param
    ops.VARIANT = old


def _csr_numpy(ei, n):
    dst = ei[1]
    order = np.argsort(dst, kind="stable")
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, dst + 1, 1)
    return np.cumsum(rowptr), order, ei[0][order], dst[order]


@pytest.mark.parametrize("case", ["davis", "knn", "ragged", "empty_edges", "isolated", "crowded", "large"])
def test_csr_from_coo(case):
    """Tables of fewer than 64k targets take the multi-workgroup scan; "large" (70k targets) the one-block tile scan."""
    if case == "davis":
        gb = ds.protein_batch(4, 3)
    elif case == "knn":
        gb = ds.protein_batch(2, 3, length=150, thresh=12, thresh_type="num")
    elif case == "ragged":
        gb = ds.protein_batch(5, 3, lengths=[1, 2, 77, 130, 3])
    ei, n = (gb.edge_index, gb.num_nodes) if case in ("davis", "knn", "ragged") else (None, None)
    if case == "empty_edges":
        ei, n = np.zeros((2, 0), np.int64), 17
    if case == "isolated":                      # nodes with no incoming edge, shuffled edge order
        rng = np.random.default_rng(0)
        n = 1000
        ei = np.stack([rng.integers(0, n, 5000), rng.integers(0, n // 2, 5000)])
    if case == "crowded":                       # ~67 edges per target
        rng = np.random.default_rng(1)
        n = 300
        ei = np.stack([rng.integers(0, n, 20000), rng.integers(0, n, 20000)])
    if case == "large":
        rng = np.random.default_rng(2)
        n = 70000
        ei = np.stack([rng.integers(0, n, 140000), rng.integers(0, n, 140000)])
    csr = ops.build_csr(torch.from_numpy(ei).to(DEV), n)
    rp, perm, src, dst = _csr_numpy(ei, n)
    E = ei.shape[1]
    assert np.array_equal(csr.rowptr.cpu().numpy(), rp)
    assert np.array_equal(csr.eperm.cpu().numpy()[:E], perm)
    assert np.array_equal(csr.esrc.cpu().numpy()[:E], src)
    assert np.array_equal(csr.edst.cpu().numpy()[:E], dst)


def _run_lba(P_state, d, num_convs=2, aggr_mean=False, stages=False):
    flat = arena.flatten_state(P_state, num_convs, DEV)
    dims = ops.make_dims()
    lay = ops.lba_layout(dims, 20, 1, num_convs)
    assert lay.total == flat.numel()
    dd = {k: (tuple(t.to(DEV) for t in v) if isinstance(v, tuple) else v.to(DEV)) for k, v in d.items()}
    csr = ops.build_csr(dd["edge_index"], dd["x"][0].shape[0])
    return ops.lba_encoder_forward(flat, lay, dims, num_convs, dd["x"][0], dd["x"][1], dd["ntypes"],
                                   dd["eattr"][0], dd["eattr"][1], dd["etypes"], csr, aggr_mean, stages)


def test_lba_golden(lba_small, protein_params):
    """The reference's own outputs (pretrained weights, ragged 3-graph batch with a
    kNN graph and zero direction vectors on self loops)."""
    g = lba_small
    T = torch.from_numpy
    d = dict(x=(T(g["x_s"]), T(g["x_v"])), edge_index=T(g["edge_index"]), ntypes=T(g["ntypes"]),
             etypes=T(g["etypes"]), eattr=(T(g["e_s"]), T(g["e_v"])))
    out, st = _run_lba(protein_params, d, stages=True)
    N = g["x_s"].shape[0]
    merged = lambda name: np.concatenate([g[f"stage_{name}_s"], g[f"stage_{name}_v"].reshape(N, -1)], 1)
    for name in ("node_embed", "conv0_dh", "conv0", "conv1_dh", "conv1"):
        assert rel_err(st[name], merged(name)) < TOL, name
    assert rel_err(out, g["out"]) < TOL
    assert rel_err(out, g["out64"]) < TOL


@pytest.mark.parametrize("shape", ["c1_davis16", "knn20", "ragged", "single_node"])
def test_lba_vs_oracle(protein_params, shape):
    if shape == "c1_davis16":                  # BASELINE config 1: 16 x 300 residues
        gb = ds.protein_batch(16, 1)
    elif shape == "knn20":                     # long-graph stress shape, ~20 edges / residue
        gb = ds.protein_batch(2, 2, length=400, thresh=20, thresh_type="num")
    elif shape == "ragged":
        gb = ds.protein_batch(6, 3, lengths=[1, 2, 65, 64, 63, 129], thresh=8.0)
    else:
        gb = ds.protein_batch(1, 4, length=1)
    d = ds.to_torch(gb)
    ref = O.protein_lba_forward(protein_params, d["x"], d["edge_index"], d["ntypes"], d["etypes"], d["eattr"])
    out = _run_lba(protein_params, d)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL


def test_lba_mean_aggr_and_depth4():
    """aggr='mean' and a CASTER-DTA(4,4)-deep stack with seeded random weights."""
    torch.manual_seed(0)
    keys = arena.lba_param_keys(4)
    dims = dict(node=(17 + 20, 3, 16, 4, 4), edge=(32 + 1, 1, 32, 1, 1), m0=(64, 9, 16, 4, 9), m=(16, 4, 16, 4, 4),
                ff0=(16, 4, 64, 8, 8), ff1=(64, 8, 16, 4, 8), head=(16, 4, 64, 0, 4))

    def gvp_shapes(si, vi, so, vo, h):
        return {"wh.weight": (h, vi), "ws.weight": (so, si + h), "ws.bias": (so,), "wv.weight": (vo, h),
                "wsv.weight": (vo, so), "wsv.bias": (vo,)}

    P = {}
    for k in keys:
        mod, leaf = k.rsplit(".", 2)[0], ".".join(k.rsplit(".", 2)[1:])
        if "scalar_norm" in k:
            n = 32 if k.startswith("gvp_edge") else 16
            P[k] = torch.rand(n) + 0.5 if k.endswith("weight") else torch.randn(n) * 0.1
            continue
        kind = ("node" if mod == "gvp_node.0" else "edge" if mod == "gvp_edge.0" else "head" if mod == "gvp_to_scalar"
                else "m0" if mod.endswith("message_func.0") else "m" if "message_func" in mod
                else "ff0" if mod.endswith("ff_func.0") else "ff1")
        shp = gvp_shapes(*dims[kind])[leaf]
        P[k] = torch.randn(*shp) * (0.3 if len(shp) > 1 else 0.1)
    gb = ds.protein_batch(3, 5, lengths=[50, 80, 33], thresh=7.0)
    d = ds.to_torch(gb)
    for mean in (False, True):
        ref = O.protein_lba_forward(P, d["x"], d["edge_index"], d["ntypes"], d["etypes"], d["eattr"],
                                    num_convs=4, aggr="mean" if mean else "sum")
        out = _run_lba(P, d, num_convs=4, aggr_mean=mean)
        assert rel_err(out, ref) < TOL, mean


def test_lba_invariances(protein_params):
    """Size-independent properties at BASELINE config-2 size (64 x 300 residues):
    rotating every vector input or permuting the edge list leaves the output
    unchanged; a batch equals its graphs run one by one."""
    gb = ds.protein_batch(64, 7)
    d = ds.to_torch(gb)
    base = _run_lba(protein_params, d)
    assert base.shape == (19200, 64) and torch.isfinite(base).all()
    q, _ = np.linalg.qr(np.random.default_rng(2).normal(size=(3, 3)))
    R = torch.from_numpy(q.astype(np.float32))
    rot = dict(d, x=(d["x"][0], d["x"][1] @ R), eattr=(d["eattr"][0], d["eattr"][1] @ R))
    assert rel_err(_run_lba(protein_params, rot), base) < TOL
    perm = torch.from_numpy(np.random.default_rng(8).permutation(gb.num_edges))
    prm = dict(d, edge_index=d["edge_index"][:, perm], etypes=d["etypes"][perm],
               eattr=(d["eattr"][0][perm], d["eattr"][1][perm]))
    assert rel_err(_run_lba(protein_params, prm), base) < TOL
    lo, hi = int(gb.ptr[5]), int(gb.ptr[6])
    em = (d["edge_index"][0] >= lo) & (d["edge_index"][0] < hi)
    one = dict(x=(d["x"][0][lo:hi], d["x"][1][lo:hi]), edge_index=d["edge_index"][:, em] - lo,
               ntypes=d["ntypes"][lo:hi], etypes=d["etypes"][em], eattr=(d["eattr"][0][em], d["eattr"][1][em]))
    assert rel_err(_run_lba(protein_params, one), base[lo:hi]) < TOL
    again = _run_lba(protein_params, d)
    assert torch.equal(again, base)            # no atomics: bitwise reproducible


def _run_gine(P, d):
    dd = {k: v.to(DEV) for k, v in d.items()}
    csr = ops.build_csr(dd["edge_index"], dd["x"].shape[0])
    x = dd["x"]
    widths = [(52, 16, 16), (16, 64, 64)]
    for l, (cin, chid, cout) in enumerate(widths):
        pf = f"conv_list.{l}."
        w = dict(eps=P[pf + "eps"], we=P[pf + "lin.weight"], be=P[pf + "lin.bias"], w0=P[pf + "nn.lins.0.weight"],
                 b0=P[pf + "nn.lins.0.bias"], w1=P[pf + "nn.lins.1.weight"], b1=P[pf + "nn.lins.1.bias"])
        w = {k: v.to(DEV) for k, v in w.items()}
        x = ops.gine_conv_forward(x, dd["ntypes"] if l == 0 else None, 11 if l == 0 else 0, dd["eattr"],
                                  dd["etypes"], 5, csr, w, cin, chid, cout, 0.01)
    return x


@pytest.mark.parametrize("n", [1, 16, 64])
def test_gine_vs_oracle(molecule_params, n):
    d = ds.to_torch(ds.drug_batch(n, 11))
    ref = O.molecule_gine_forward(molecule_params, d["x"], d["edge_index"], d["ntypes"], d["etypes"], d["eattr"])
    out = _run_gine(molecule_params, d)
    assert out.shape == ref.shape and rel_err(out, ref) < TOL


def test_unsupported_dims_fail_loudly(protein_params):
    gb = ds.protein_batch(1, 0, length=20)
    d = ds.to_torch(gb)
    bad = dict(d, x=(d["x"][0][:, :16], d["x"][1]))
    with pytest.raises(NotImplementedError):
        _run_lba(protein_params, bad)
    with pytest.raises(RuntimeError):
        ops.build_csr(d["edge_index"], gb.num_nodes)      # CPU tensor: no CPU path


def test_csr_collate_equals_from_coo():
    """Batch CSR assembled from per-graph CSRs (one launch) == CSR built from the concatenated edge_index."""
    gb = ds.protein_batch(7, 21, lengths=[1, 40, 300, 2, 129, 64, 33])
    ptr, eptr = [int(v) for v in gb.ptr], [int(v) for v in gb.eptr]
    ei = torch.as_tensor(gb.edge_index, dtype=torch.int64)
    graphs = [(ei[:, eptr[g]:eptr[g + 1]] - ptr[g]).contiguous().to(DEV) for g in range(gb.num_graphs)]
    sizes = [ptr[g + 1] - ptr[g] for g in range(gb.num_graphs)]
    store = ops.CsrStore(graphs, sizes)
    order = [3, 0, 6, 2, 2, 5, 1, 4]                       # a batch may repeat a graph and reorder them
    batch_ei = torch.cat([graphs[g] + off for g, off in zip(order, [0] + list(torch.tensor([sizes[g] for g in order]).cumsum(0)[:-1]))], dim=1)
    plan = store.plan(order)
    got = store.collate(plan, attach_to=batch_ei)
    ref = ops.build_csr(batch_ei, plan["N"])
    assert plan["E"] == ref.num_edges
    for name in ("rowptr", "eperm", "esrc", "edst"):
        a, b = getattr(got, name), getattr(ref, name)
        n = plan["N"] + 1 if name == "rowptr" else plan["E"]
        assert torch.equal(a[:n].cpu(), b[:n].cpu()), name
    assert ops.cached_csr(batch_ei, plan["N"]) is got      # the encoders pick the attached tables up


def test_feature_table_wire_format(protein_params):
    """unique graphs' edge features stay resident on the device in dst-sorted order; a batch is a list of
    graph ids.  Running the encoder on (store tables, eperm into the tables) == running it on the batch collated the
    reference's way (concatenated features + COO sort), forward and every gradient."""
    import json, os
    from conftest import GOLDEN
    from models.protein_gnn import SelectableProteinModelWrapper
    if ops.VARIANT != "mfma":
        pytest.skip("training path: MFMA kernels")
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["protein_gnn_kwargs"]
    for k in ("in_channels", "edge_dim", "hidden_channels", "edge_hidden_channels"):
        kw[k] = tuple(kw[k])
    model = SelectableProteinModelWrapper(**kw)
    model.load_state_dict({"gnn_model." + k: v for k, v in protein_params.items()})
    model = model.to(DEV).eval()
    rng = np.random.default_rng(12)
    graphs = [ds.protein_graph(L, rng, 4.0, "dist") for L in (40, 77, 33, 120)]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    store = ops.CsrStore([T(g["edge_index"]) for g in graphs], [g["x_s"].shape[0] for g in graphs],
                         edge_attr=[(T(g["e_s"]), T(g["e_v"])) for g in graphs], edge_types=[T(g["etypes"]) for g in graphs])
    order = [2, 0, 3, 3, 1]                                   # a batch may repeat a graph and reorder them
    gb = ds.collate([graphs[i] for i in order])
    d = {k: (tuple(t.to(DEV) for t in v) if isinstance(v, tuple) else v.to(DEV)) for k, v in ds.to_torch(gb).items()}
    params = [p for p in model.parameters() if p.numel()]
    r = torch.randn(gb.num_nodes, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    ref = model(d["x"], d["edge_index"], d["ntypes"], d["etypes"], eattr=d["eattr"])
    g_ref = torch.autograd.grad(ref, params, r)
    plan = store.plan(order)
    ei = d["edge_index"].clone()                              # a fresh tensor object: carries only the attached tables
    csr = store.collate(plan, attach_to=ei, table=True)
    assert csr.table_rows == sum(g["e_s"].shape[0] for g in graphs) and plan["E"] == gb.num_edges
    eattr, etypes = store.edge_table()
    old = ops.CSR_CACHE_ENABLED
    ops.CSR_CACHE_ENABLED = True
    try:
        out = model(d["x"], ei, d["ntypes"], etypes, eattr=eattr)
        g_out = torch.autograd.grad(out, params, r)
    finally:
        ops.CSR_CACHE_ENABLED = old
    assert torch.equal(out, ref)                              # same kernels, same per-edge arithmetic, same order
    scale = max(float(b.abs().max()) for b in g_ref)           # (analytically-zero gradients are rounding noise that the
    for a, b in zip(g_out, g_ref):                              # backward's d h[src] atomics reorder: floor tied to the scale)
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-7 * scale


def test_csr_build_with_dirty_counters_stays_inside_its_tables():
    """The CSR build's contract is "counters zero on entry".  Round 2's recorded GPU fault was what a violated contract
    did then: un-zeroed counters (a captured hipMemsetAsync that re-zeroed only part of them on graph replays) sent
    csr_fill's stores past the E-element tables.  The kernels now bound every access by E / N: with counters full of
    garbage the tables are wrong, but nothing outside them is touched (sentinel guards on both sides of every table)
    and the counters come back zeroed; the next build on clean counters is correct again."""
    import ctypes as C
    from gvp_hip import _lib
    gb = ds.protein_batch(3, 5, lengths=[40, 77, 25])
    ei = torch.from_numpy(gb.edge_index).to(DEV)
    N, E = gb.num_nodes, gb.num_edges
    G = 4096                                                     # guard ints around every table
    arena = torch.full((5 * (E + 2 * G) + N + 1 + 2 * G,), -7, dtype=torch.int32, device=DEV)
    tabs, off = [], 0
    for n in (N + 1, E, E, E, E):
        tabs.append(arena[off + G:off + G + n])
        off += n + 2 * G
    rowptr, eperm, esrc, edst, ids = tabs
    L = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    counters = torch.zeros((N + 1 + 63) // 64 * 64, dtype=torch.int32, device=DEV)
    for dirty in (1000, 2 ** 30, -5):
        counters.fill_(dirty)
        before = arena.clone()
        for t in tabs:
            t.fill_(-7)
        before = arena.clone()
        rc = L.cgvp_csr_from_coo(P(ei), N, E, P(rowptr), P(eperm), P(esrc), P(edst), P(counters), 1, P(ids), st)
        torch.cuda.synchronize()
        assert rc == 0
        mask = torch.ones_like(arena, dtype=torch.bool)
        for t in tabs:
            lo = (t.data_ptr() - arena.data_ptr()) // 4
            mask[lo:lo + t.numel()] = False
        assert torch.equal(arena[mask], before[mask]), dirty        # every guard int untouched
        assert int(counters.abs().max()) == 0                       # and the contract is restored for the next call
    rc = L.cgvp_csr_from_coo(P(ei), N, E, P(rowptr), P(eperm), P(esrc), P(edst), P(counters), 1, P(ids), st)
    torch.cuda.synchronize()
    order = np.lexsort((np.arange(E), gb.edge_index[1]))           # stable by target, then edge id
    assert np.array_equal(eperm.cpu().numpy(), order)
