"""GPU parity at the drop-in boundary: the `models.*` nn.Modules (what
train_model.py / inference_utils.py instantiate) against the reference's golden
vectors and the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

import davis_synth as ds
from conftest import GOLDEN, rel_err
from oracle import gvp_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-5


def _to(d, dev=DEV):
    return {k: (tuple(t.to(dev) for t in v) if isinstance(v, tuple) else v.to(dev)) for k, v in d.items()}


@pytest.fixture(scope="module")
def joint(pretrained):
    from models.joint_gnn import JointGNN
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
    m = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"],
                 **kw["joint_gnn_kwargs"])
    m.load_state_dict(pretrained, strict=True)
    return m.to(DEV).eval()


def test_protein_module_golden(joint, lba_small):
    g = lba_small
    T = torch.from_numpy
    d = dict(x=(T(g["x_s"]), T(g["x_v"])), edge_index=T(g["edge_index"]), ntypes=T(g["ntypes"]),
             etypes=T(g["etypes"]), eattr=(T(g["e_s"]), T(g["e_v"])), batch=T(g["batch"]))
    with torch.no_grad():
        out = joint.protein_gnn(**_to(d))
    assert rel_err(out, g["out"]) < TOL


def test_joint_model_vs_oracle(joint, pretrained):
    """Encoders (HIP) + head (stock torch on the GPU) == oracle end to end, on a
    ragged batch; exercises to_dense_batch padding and the attention masks."""
    p, m = ds.pair_batch(5, 21, lengths=[40, 75, 33, 120, 64])
    pd, md = ds.to_torch(p), ds.to_torch(m)
    with torch.no_grad():
        y, attn = joint(_to(pd), _to(md))
        atoms = joint.molecule_gnn(**_to(md))
    ref = O.joint_forward(pretrained, pd, md)
    mp = {k[len("molecule_gnn.gnn_model."):]: v for k, v in pretrained.items() if k.startswith("molecule_gnn.gnn_model.")}
    ref_atoms = O.molecule_gine_forward(mp, md["x"], md["edge_index"], md["ntypes"], md["etypes"], md["eattr"])
    assert rel_err(atoms, ref_atoms) < TOL
    assert y.shape == (5, 1) and rel_err(y, ref) < 1e-4
    assert attn[0][0].shape[0] == 5


def test_state_dict_roundtrip_and_arena_refresh(joint, pretrained):
    """Weights changed through the ordinary nn.Module API reach the kernels."""
    p = ds.protein_batch(2, 3, lengths=[50, 31])
    d = _to(ds.to_torch(p))
    with torch.no_grad():
        base = joint.protein_gnn(**d).clone()
        w = joint.protein_gnn.gnn_model.gvp_to_scalar.ws.bias
        w.add_(0.5)
        bumped = joint.protein_gnn(**d)
        assert float((bumped - base).abs().max()) > 0.1
        joint.load_state_dict(pretrained, strict=True)
        assert torch.equal(joint.protein_gnn(**d), base)
        sd = {k: v.clone() for k, v in joint.state_dict().items()}
    assert all(torch.equal(sd[k].cpu(), pretrained[k]) for k in pretrained)


def test_csr_memo_semantics(joint):
    from gvp_hip import ops
    p = ds.protein_batch(1, 5, length=40)
    d = _to(ds.to_torch(p))
    ei = d["edge_index"]
    a = ops.cached_csr(ei, p.num_nodes)
    assert ops.cached_csr(ei, p.num_nodes) is a
    assert ops.cached_csr(ei.clone(), p.num_nodes) is not a      # new tensor object: rebuilt
    ei[0, 0] = ei[0, 0]                                          # in-place write bumps _version
    assert ops.cached_csr(ei, p.num_nodes) is not a


def test_hip_graph_capture_replays(joint):
    """The whole encoder step (CSR build included) is capturable in a HIP graph."""
    from gvp_hip import ops
    p, m = ds.pair_batch(4, 9, length=80)
    pd, md = _to(ds.to_torch(p)), _to(ds.to_torch(m))
    old = ops.CSR_CACHE_ENABLED
    ops.CSR_CACHE_ENABLED = False
    try:
        with torch.no_grad():
            eager = joint.protein_gnn(**pd).clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                joint.protein_gnn(**pd)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = joint.protein_gnn(**pd)
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, eager)
    finally:
        ops.CSR_CACHE_ENABLED = old


def test_pair_parallel_world1_on_the_hip_encoders(pretrained):
    """`enable_pair_parallel()` on the real encoders with RCCL at world_size 1 (`always_communicate=True`: the
    all-gather and the flat gradient all-reduce are issued although one rank needs neither): same prediction and same
    gradients as the plain model.  The multi-rank arithmetic is covered by tests/test_pair_parallel_cpu.py (gloo)."""
    import socket
    import torch.distributed as dist
    from models.joint_gnn import JointGNN
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))

    def build():
        m = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"],
                     **kw["joint_gnn_kwargs"])
        m.load_state_dict(pretrained, strict=True)
        return m.to(DEV).eval()

    p, m = ds.pair_batch(6, 5, lengths=[40, 75, 33, 120, 64, 51])
    pd, md = _to(ds.to_torch(p)), _to(ds.to_torch(m))
    plain = build()
    y0, _ = plain(pd, md)
    y0.square().sum().backward()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        par = build().enable_pair_parallel(pair_counts=[6], always_communicate=True)
        y1, _ = par(pd, md)
        y1.square().sum().backward()
        par.reduce_pair_parallel_grads()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert rel_err(y1, y0) < 1e-6
    g0, g1 = dict(plain.named_parameters()), dict(par.named_parameters())
    scale = max(float(v.grad.abs().max()) for v in g0.values() if v.grad is not None)
    n = 0
    for k, v in g0.items():
        if v.grad is None:
            continue
        assert float((g1[k].grad - v.grad).abs().max()) <= 1e-5 * float(v.grad.abs().max()) + 2e-6 * scale, k
        n += 1
    assert n > 100


def test_fragment_image_cache_is_per_model(lba_small):
    """Two freshly built models one after the other (the second one's parameter arena can land on the freed first one's
    address, with equal version counters): each must run on ITS OWN weights, in the autograd path too."""
    from models.protein_gnn import SelectableProteinModelWrapper
    import gc
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))["protein_gnn_kwargs"]
    for k in ("in_channels", "edge_dim", "hidden_channels", "edge_hidden_channels"):
        kw[k] = tuple(kw[k])
    g = lba_small
    T = torch.from_numpy
    d = _to(dict(x=(T(g["x_s"]), T(g["x_v"])), edge_index=T(g["edge_index"]), ntypes=T(g["ntypes"]),
                 etypes=T(g["etypes"]), eattr=(T(g["e_s"]), T(g["e_v"]))))
    for seed in (1, 2, 3):
        torch.manual_seed(seed)
        m = SelectableProteinModelWrapper(**kw).to(DEV).eval()
        P = {k: v.detach().cpu() for k, v in m.gnn_model.state_dict().items()}
        ref = O.protein_lba_forward(P, (T(g["x_s"]), T(g["x_v"])), T(g["edge_index"]), T(g["ntypes"]), T(g["etypes"]),
                                    (T(g["e_s"]), T(g["e_v"])))
        xs = d["x"][0].clone().requires_grad_()                       # autograd path (the cached-image path)
        out = m((xs, d["x"][1]), d["edge_index"], d["ntypes"], d["etypes"], eattr=d["eattr"])
        assert rel_err(out, ref) < TOL, seed
        with torch.no_grad():
            assert rel_err(m(**d), ref) < TOL, seed
        del m, out, xs
        gc.collect()
        torch.cuda.empty_cache()


def test_joint_training_step_replays_from_a_hip_graph(pretrained):
    """The whole JointGNN training step (both encoders, varlen attention, torch head, backward of everything) captured
    into one HIP graph: replays reproduce the eager gradients (dropout switched off to compare), twice in a row, and a
    weight update between replays is seen by the next replay (the weight image is rebuilt inside the graph).
    (This test found that hipMemsetAsync, captured with a size that is not a multiple of 256 B, leaves the buffer
    un-zeroed from the second replay on: the library now zero-fills with its own kernel.)"""
    from models.joint_gnn import JointGNN
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
    kw["protein_gnn_kwargs"] = dict(kw["protein_gnn_kwargs"], dropout_rate=0.0)
    kw["molecule_gnn_kwargs"] = dict(kw["molecule_gnn_kwargs"], dropout_rate=0.0)
    model = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"],
                     **kw["joint_gnn_kwargs"])
    model.load_state_dict(pretrained, strict=True)
    model = model.to(DEV).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    p, m = ds.pair_batch(6, 9, lengths=[40, 75, 33, 120, 64, 51])
    pd, md = _to(ds.to_torch(p)), _to(ds.to_torch(m))
    pd["ptr"], md["ptr"] = torch.as_tensor(p.ptr).to(DEV), torch.as_tensor(m.ptr).to(DEV)
    params = [q for q in model.parameters() if q.numel()]
    target = torch.randn(6, 1, device=DEV, generator=torch.Generator(device=DEV).manual_seed(0))

    def step():
        pred, _ = model(pd, md)
        return torch.autograd.grad(torch.nn.functional.mse_loss(pred, target), params)

    eager = [g.clone() for g in step()]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        out = step()
    scale = max(float(g.abs().max()) for g in eager)

    names = [n for n, q in model.named_parameters() if q.numel()]

    def worst(a, b):
        errs = [(float((x - y).abs().max()) / (1e-4 * float(y.abs().max()) + 1e-5 * scale), n, float(x.abs().max()), float(y.abs().max()))
                for x, y, n in zip(a, b, names)]
        return sorted(errs, reverse=True)[:4]

    def same(a, b):
        return worst(a, b)[0][0] <= 1.0

    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        assert same(out, eager), worst(out, eager)
    with torch.no_grad():                                            # an "optimizer step": the next replay must see it
        model.protein_gnn.gnn_model.gvp_to_scalar.ws.weight.mul_(1.5)
        model.output_layer.weight.mul_(0.5)
    graph.replay()
    torch.cuda.synchronize()
    captured = [g.clone() for g in out]
    assert not same(captured, eager)
    assert same(captured, step())                                    # eager on the updated weights


def test_bench_style_two_stream_step_replays_correctly(pretrained):
    """What bench.py times: protein forward + backward on the capture stream, drug forward + backward on a side stream,
    CSR tables rebuilt inside the step, all of it one HIP graph.  Three replays in a row return the eager gradients of
    both encoders (dropout off to compare)."""
    from gvp_hip import ops
    from models.joint_gnn import JointGNN
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
    kw["protein_gnn_kwargs"] = dict(kw["protein_gnn_kwargs"], dropout_rate=0.0)
    kw["molecule_gnn_kwargs"] = dict(kw["molecule_gnn_kwargs"], dropout_rate=0.0)
    model = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"],
                     **kw["joint_gnn_kwargs"])
    model.load_state_dict(pretrained, strict=True)
    model = model.to(DEV).train()
    p, m = ds.pair_batch(8, 4, lengths=[40, 75, 33, 120, 64, 51, 90, 17])
    pd, md = _to(ds.to_torch(p)), _to(ds.to_torch(m))
    pp = [q for q in model.protein_gnn.parameters() if q.numel()]
    mp = [q for q in model.molecule_gnn.parameters() if q.numel()]
    gen = torch.Generator(device=DEV).manual_seed(2)
    g_res = torch.randn(p.num_nodes, 64, device=DEV, generator=gen)
    g_atm = torch.randn(m.num_nodes, 64, device=DEV, generator=gen)
    side = torch.cuda.Stream()

    def step():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        res = model.protein_gnn(**pd)
        with torch.cuda.stream(side):
            atm = model.molecule_gnn(**md)
        gp = torch.autograd.grad([res], pp, [g_res])
        with torch.cuda.stream(side):
            gd = torch.autograd.grad([atm], mp, [g_atm])
        main.wait_stream(side)
        return gp + gd

    old = ops.CSR_CACHE_ENABLED
    ops.CSR_CACHE_ENABLED = False
    try:
        eager = [g.clone() for g in step()]
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            out = step()
        scale = max(float(g.abs().max()) for g in eager)
        for rep in range(3):
            graph.replay()
            torch.cuda.synchronize()
            for a, b in zip(out, eager):
                assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 2e-6 * scale, rep
    finally:
        ops.CSR_CACHE_ENABLED = old


def test_captured_step_survives_regrow_reseed_and_frees_with_dropout_on(pretrained):
    """Lifetime of what a captured step writes to (the hazards named there: the CSR counters were
    re-allocated when a larger batch arrived, the generator state was re-drawn after torch.manual_seed, cached images
    were evicted -- each a use-after-free for a live graph).  Now: both workspaces of a pass come from the graph's own
    pool, the fragment image lives in the forward workspace, the counters are never freed (a larger batch appends a
    generation), the generator state is re-seeded IN PLACE.
    The bench-style two-stream training step of both encoders at davis_b64 size, DROPOUT ON, is captured; then, on the
    capture stream, a larger batch runs eagerly (counters grow), torch.manual_seed re-seeds, junk is allocated and freed
    and the allocator's cache is emptied; the graph is replayed three times: finite, different masks every replay, and
    every gradient equals an EAGER step run with the replay's own {seed, offset}."""
    from gvp_hip import autograd_ops, ops
    from models.joint_gnn import JointGNN
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
    model = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"],
                     **kw["joint_gnn_kwargs"])
    model.load_state_dict(pretrained, strict=True)
    model = model.to(DEV).train()
    p, m = ds.pair_batch(64, 0)                                   # davis_b64: 19,200 residues, ~57k edges
    big_p, big_m = ds.pair_batch(130, 1, length=320)              # 41,600 residues: more than the counters were sized for (2 x 19,264)
    pd, md = _to(ds.to_torch(p)), _to(ds.to_torch(m))
    bpd, bmd = _to(ds.to_torch(big_p)), _to(ds.to_torch(big_m))
    pp = [q for q in model.protein_gnn.parameters() if q.numel()]
    mp = [q for q in model.molecule_gnn.parameters() if q.numel()]
    gen = torch.Generator(device=DEV).manual_seed(2)
    g_res = torch.randn(p.num_nodes, 64, device=DEV, generator=gen)
    g_atm = torch.randn(m.num_nodes, 64, device=DEV, generator=gen)
    side = torch.cuda.Stream()

    def step(a=pd, b=md, gr=g_res, ga=g_atm):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        res = model.protein_gnn(**a)
        with torch.cuda.stream(side):
            atm = model.molecule_gnn(**b)
        gp = torch.autograd.grad([res], pp, [gr])
        with torch.cuda.stream(side):
            gd = torch.autograd.grad([atm], mp, [ga])
        main.wait_stream(side)
        return gp + gd

    old = ops.CSR_CACHE_ENABLED
    ops.CSR_CACHE_ENABLED = False
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        counters_before = {k: [t.data_ptr() for t in v] for k, v in ops._COUNTERS.items()}
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            out = step()
        ws_lba, ws_gine = autograd_ops._LAST_WS["lba"], autograd_ops._LAST_WS["gine"]      # the graph's own workspaces
        seed_of = lambda w: w[0][w[1]:w[1] + 16].view(torch.int64)
        graph.replay()
        torch.cuda.synchronize()
        first = [g.clone() for g in out]
        # ---- the hazards, all on the capture stream
        with torch.cuda.stream(s):
            gr2 = torch.randn(big_p.num_nodes, 64, device=DEV)
            ga2 = torch.randn(big_m.num_nodes, 64, device=DEV)
            step(bpd, bmd, gr2, ga2)                               # larger batch: the per-stream counters grow
            torch.manual_seed(4242)                                # re-seed: in place
            step(bpd, bmd, gr2, ga2)
            junk = [torch.empty(1 << 22, device=DEV).normal_() for _ in range(8)]
            del junk
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        grown = any(len(v) > len(counters_before.get(k, [])) for k, v in ops._COUNTERS.items())
        assert grown                                               # a new generation was appended ...
        for k, ptrs in counters_before.items():                    # ... and every old buffer is still alive, same address
            assert [t.data_ptr() for t in ops._COUNTERS[k]][:len(ptrs)] == ptrs
        seeds = []
        for rep in range(3):
            graph.replay()
            torch.cuda.synchronize()
            assert all(torch.isfinite(g).all() for g in out), rep
            seeds.append((seed_of(ws_lba).clone(), seed_of(ws_gine).clone()))
        assert int(seeds[2][0][1]) == int(seeds[1][0][1]) + 1      # the generator advances on every replay
        replayed = [g.clone() for g in out]
        assert not all(torch.equal(a, b) for a, b in zip(first, replayed))      # other masks than the first replay
        # ---- the last replay against an eager step with the same {seed, offset}
        for kind, sd in (("lba", seeds[2][0]), ("gine", seeds[2][1])):
            st = autograd_ops.rng_state(kind, torch.device(DEV))
            st.copy_(torch.stack([sd[0], sd[1] - 1]))
        eager = step()
        torch.cuda.synchronize()
        assert torch.equal(autograd_ops.last_seed("lba"), seeds[2][0]) and torch.equal(autograd_ops.last_seed("gine"), seeds[2][1])
        scale = max(float(g.abs().max()) for g in eager)
        for a, b in zip(replayed, eager):
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 2e-6 * scale
    finally:
        ops.CSR_CACHE_ENABLED = old


def test_bucketed_graph_replay_equals_eager_on_changing_batches(pretrained):
    """gvp_hip.graphed.GraphedEncoderStep: a different batch every step (other proteins, other N / E), replayed from one
    captured step per shape bucket with the batch staged into padded static buffers (padded residues / atoms isolated,
    padded edges with endpoint -1).  Embeddings of the real rows and every weight gradient equal the eager step on the
    unpadded batch; batches of different sizes share a bucket when their rounded sizes agree."""
    from gvp_hip.graphed import GraphedEncoderStep, bucket_size
    from models.joint_gnn import JointGNN
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
    kw["protein_gnn_kwargs"] = dict(kw["protein_gnn_kwargs"], dropout_rate=0.0)
    kw["molecule_gnn_kwargs"] = dict(kw["molecule_gnn_kwargs"], dropout_rate=0.0)
    model = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"],
                     **kw["joint_gnn_kwargs"])
    model.load_state_dict(pretrained, strict=True)
    model = model.to(DEV).train()
    runner = GraphedEncoderStep(model.protein_gnn, model.molecule_gnn)
    pp, mp = runner.pp, runner.mp
    cases = [(3, [40, 75, 33, 120, 64, 51]), (4, [41, 74, 33, 121, 64, 50]), (5, [300, 280]), (6, [39, 76, 33, 119, 65, 51]),
             (7, [17])]
    gen = torch.Generator(device=DEV).manual_seed(0)
    seen = set()
    for seed, lengths in cases:
        p, m = ds.pair_batch(len(lengths), seed, lengths=lengths)
        pd, md = _to(ds.to_torch(p)), _to(ds.to_torch(m))
        pd.pop("batch"), md.pop("batch")
        g_res = torch.randn(p.num_nodes, 64, device=DEV, generator=gen)
        g_atm = torch.randn(m.num_nodes, 64, device=DEV, generator=gen)
        res, atm, gp, gd = runner.run(pd, md, g_res, g_atm)
        got = [t.clone() for t in (res, atm) + tuple(gp) + tuple(gd)]
        seen.add((bucket_size(p.num_nodes), bucket_size(p.num_edges), bucket_size(m.num_nodes), bucket_size(m.num_edges)))
        e_res = model.protein_gnn(**pd)
        e_atm = model.molecule_gnn(**md)
        want = [e_res.detach(), e_atm.detach()] + list(torch.autograd.grad([e_res, e_atm], pp + mp, [g_res, g_atm]))
        assert got[0].shape == want[0].shape and got[1].shape == want[1].shape
        assert rel_err(got[0], want[0]) < 1e-6 and rel_err(got[1], want[1]) < 1e-6
        scale = max(float(w.abs().max()) for w in want[2:])
        for a, b in zip(got[2:], want[2:]):
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 2e-6 * scale, (seed, tuple(a.shape))
    assert len(runner.buckets) == len(seen) < len(cases)               # cases 1, 2 and 4 share a bucket


def test_fused_parameter_mode_trains_like_the_per_tensor_model(pretrained, lba_small, monkeypatch):
    """`JointGNN.fuse_encoder_parameters()`: one autograd leaf for the protein encoder.  Same outputs, the gradient of
    the arena is the per-tensor gradients laid end to end (against the REFERENCE's autograd numbers of lba_small too),
    and two SGD-with-momentum steps move both models to the same weights (checkpoints compared key by key).  Eager C++ fast path
    and the Python custom-op path."""
    from models.joint_gnn import JointGNN
    from gvp_hip.arena import lba_param_keys
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
    make = lambda: JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"],
                            **kw["joint_gnn_kwargs"])
    plain, fused = make(), make()
    for m in (plain, fused):
        m.load_state_dict(pretrained, strict=True)
        m.to(DEV).train()
    fused.fuse_encoder_parameters()
    gm = fused.protein_gnn.gnn_model
    g = lba_small
    T = torch.from_numpy
    d = _to(dict(x=(T(g["x_s"]), T(g["x_v"])), edge_index=T(g["edge_index"]), ntypes=T(g["ntypes"]),
                 etypes=T(g["etypes"]), eattr=(T(g["e_s"]), T(g["e_v"])), batch=T(g["batch"])))
    r = T(g["r"]).to(DEV)
    from gvp_hip import autograd_ops
    for use_bridge in (True, False):
        with monkeypatch.context() as mp:
            if not use_bridge:                               # the torch.library custom ops in eager mode
                mp.setattr(autograd_ops, "_eager_bridge", lambda: None)
            for m in (plain, fused):
                m.eval()
                m.zero_grad(set_to_none=True)
            out_p = plain.protein_gnn(**d)
            out_f = fused.protein_gnn(**d)
            assert rel_err(out_f, g["out"]) < TOL and rel_err(out_f, out_p.detach().cpu().numpy()) < 1e-6
            (out_p * r).sum().backward()
            (out_f * r).sum().backward()
            named = dict(plain.protein_gnn.gnn_model.named_parameters())
            flat = torch.cat([named[k].grad.reshape(-1) for k in lba_param_keys(2) if named[k].numel()])
            assert gm.arena.grad.shape == (15117,)
            assert float((gm.arena.grad - flat).abs().max()) <= 1e-5 * float(flat.abs().max())
            off = 0
            for k in lba_param_keys(2):                      # ... and against the reference's autograd
                n = named[k].numel()
                if n:
                    ref = T(g["g_" + k]).to(DEV).reshape(-1)
                    assert float((gm.arena.grad[off:off + n] - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 2e-6 * float(flat.abs().max()), k
                    off += n
    # drug encoder: the arena's gradient through the C++ node (one leaf) and through the custom op (connected views)
    # equals the per-tensor gradients of the unfused model laid end to end
    p6, m6 = ds.pair_batch(6, 3)
    md = _to(ds.to_torch(m6))
    dm = fused.molecule_gnn.gnn_model
    ga = torch.randn(md["x"].shape[0], 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    plain.zero_grad(set_to_none=True)
    (plain.molecule_gnn(**md) * ga).sum().backward()
    keys = ("eps", "nn.lins.0.weight", "nn.lins.0.bias", "nn.lins.1.weight", "nn.lins.1.bias", "lin.weight", "lin.bias")
    named = dict(plain.molecule_gnn.gnn_model.named_parameters())
    dflat = torch.cat([named[f"conv_list.{l}.{k}"].grad.reshape(-1) for l in range(2) for k in keys])
    for use_bridge in (True, False):
        with monkeypatch.context() as mp:
            if not use_bridge:
                mp.setattr(autograd_ops, "_eager_bridge", lambda: None)
            fused.zero_grad(set_to_none=True)
            (fused.molecule_gnn(**md) * ga).sum().backward()
            assert dm.arena.grad.shape == (7390,)
            assert float((dm.arena.grad - dflat).abs().max()) <= 1e-5 * float(dflat.abs().max()), use_bridge
    # two optimizer steps (dropout off so both models see the same numbers)
    # (SGD: linear in the gradient.  Adam's first steps move every weight by lr * sign(g), which turns the run-to-run
    # rounding noise of analytically-zero gradients -- float atomics in d h[src] -- into +-lr differences)
    opts = [torch.optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-2) for m in (plain, fused)]
    pd = _to(ds.to_torch(p6))
    for _ in range(2):
        for m, opt in zip((plain, fused), opts):
            m.eval()
            opt.zero_grad(set_to_none=True)
            y, _ = m(pd, md)
            y.square().sum().backward()
            opt.step()
    sp, sf = plain.state_dict(), fused.state_dict()
    assert set(sp) == set(sf)
    moved = 0
    for k in sp:
        if not sp[k].numel():                                # the zero-size dummy_params
            continue
        assert float((sp[k] - sf[k]).abs().max()) <= 2e-5 * max(float(sp[k].abs().max()), 1e-3), k
        moved += int(not torch.equal(sp[k].cpu(), pretrained[k]))
    assert moved > 100


def test_two_lane_head_equals_the_single_stream_head(pretrained):
    """`model.two_stream_head = True` (opt-in): the atom side of the head on a side stream.  Same predictions and the same
    764,396 gradients as the single-stream head, eagerly (3 steps back to back, so that a missing cross-stream
    dependency would show) and from a captured graph."""
    from models.joint_gnn import JointGNN
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
    model = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"],
                     **kw["joint_gnn_kwargs"])
    model.load_state_dict(pretrained, strict=True)
    model.to(DEV).eval()                                   # eval: no dropout, the two runs see the same numbers
    model.attention_weights = "never"                      # (the padded weight tensors need a host-side maximum: not capturable)
    p, m = ds.pair_batch(24, 11)
    pd, md = _to(ds.to_torch(p)), _to(ds.to_torch(m))
    pd["ptr"], md["ptr"] = torch.as_tensor(p.ptr).to(DEV), torch.as_tensor(m.ptr).to(DEV)
    params = [q for q in model.parameters() if q.numel()]
    target = torch.randn(24, 1, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))

    def step():
        pred, _ = model(pd, md)
        loss = torch.nn.functional.mse_loss(pred, target)
        return [pred.detach().clone()] + [g.detach().clone() for g in torch.autograd.grad(loss, params)]

    model.two_stream_head = False
    ref = step()
    model.two_stream_head = True
    for _ in range(3):
        got = step()
    torch.cuda.synchronize()
    scale = max(float(g.abs().max()) for g in ref[1:])
    for a, b in zip(got, ref):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 2e-7 * scale
    # training mode with every dropout on: the masks are pure functions of the generator states, the call site and the
    # element, so with the states restored the two schedules must agree here too
    from gvp_hip import autograd_ops, head_ops
    model.train()
    model.two_stream_head = False
    step()                                                  # (creates the three device-side generator states)
    dev = torch.device(DEV)
    states = [autograd_ops.rng_state("lba", dev), autograd_ops.rng_state("gine", dev), head_ops._head_rng_state(dev)]
    saved = [t.clone() for t in states]
    ref_t = step()
    for t, v in zip(states, saved):
        t.copy_(v)
    model.two_stream_head = True
    got_t = step()
    torch.cuda.synchronize()
    scale_t = max(float(g.abs().max()) for g in ref_t[1:])
    for a, b in zip(got_t, ref_t):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 2e-7 * scale_t
    assert not torch.allclose(ref_t[0], ref[0])            # (dropout did act)
    model.eval()
    # captured: the side lane becomes a parallel branch of the graph
    s = torch.cuda.Stream(device=DEV)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
        torch.cuda.current_stream().synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = step()
        for _ in range(3):
            g.replay()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    for a, b in zip(out, ref):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 2e-7 * scale


def test_two_lane_head_without_cross_attention_and_without_ptr(pretrained):
    """with `cross_attn_module = None` and no 'ptr' in the dicts, the atom-side offsets are created on the launch
    stream and read by the pooling on the atom lane: the lane has to be forked after they exist.  Same outputs and gradients
    as the single-stream head, several steps back to back."""
    from models.joint_gnn import JointGNN
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
    jk = dict(kw["joint_gnn_kwargs"])
    model = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"], **jk)
    model.load_state_dict(pretrained, strict=True)
    model.cross_attn_module = None
    model.to(DEV).eval()
    p, m = ds.pair_batch(24, 12)
    pd, md = _to(ds.to_torch(p)), _to(ds.to_torch(m))
    pd.pop("ptr", None), md.pop("ptr", None)
    params = [q for q in model.parameters() if q.numel()]
    target = torch.randn(24, 1, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))

    def step():
        pred, _ = model(pd, md)
        loss = torch.nn.functional.mse_loss(pred, target)
        return [pred.detach().clone()] + [g.detach().clone() for g in torch.autograd.grad(loss, params, allow_unused=True) if g is not None]

    model.two_stream_head = False
    ref = step()
    model.two_stream_head = True
    for _ in range(4):
        got = step()
    torch.cuda.synchronize()
    scale = max(float(g.abs().max()) for g in ref[1:])
    assert len(got) == len(ref)
    for a, b in zip(got, ref):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 2e-7 * scale


def test_backward_writes_leaf_grads_without_engine_tasks_and_keeps_autograd_semantics(pretrained):
    """Round 4: a plain `loss.backward()` lets each encoder's node write its parameters' .grad itself (LeafScatter in
    csrc/torch_bridge.cpp) instead of handing the engine one task per leaf.  Everything autograd promises must still hold:
    same gradients as the stock path, accumulation into existing .grad, frozen parameters, torch.autograd.grad and
    backward(inputs=...) on the same graph, tensor hooks and post-accumulate hooks (which switch that pass to the stock path)."""
    from gvp_hip import _lib
    from models.joint_gnn import JointGNN
    bridge = _lib.bridge()
    if bridge is None:
        pytest.skip("C++ bridge not in use")
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
    model = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"],
                     **kw["joint_gnn_kwargs"])
    model.load_state_dict(pretrained, strict=True)
    model.to(DEV).eval()
    p, m = ds.pair_batch(6, 21)
    pd, md = _to(ds.to_torch(p)), _to(ds.to_torch(m))
    enc = [(n, q) for n, q in model.named_parameters() if q.numel() and (n.startswith("protein_gnn") or n.startswith("molecule_gnn"))]
    params = [q for _, q in enc]
    target = torch.randn(6, 1, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))

    gscale = [1.0]

    def same(x, y):            # (the backward's d h[src] float atomics make two runs differ in the last bits; analytically-zero
        return float((x - y).abs().max()) <= 2e-5 * float(y.abs().max()) + 1e-6 * gscale[0]      # gradients are pure noise)

    def loss_of():
        pred, _ = model(pd, md)
        return torch.nn.functional.mse_loss(pred, target)

    def grads_after_backward(scale=1.0):
        (loss_of() * scale).backward()
        return [q.grad.detach().clone() for q in params]

    was = bridge.set_exact_leaves(True)
    try:
        model.zero_grad(set_to_none=True)
        ref = grads_after_backward()
        gscale[0] = max(float(g.abs().max()) for g in ref)
        ref2 = grads_after_backward(0.5)                       # accumulated on top, the stock way
        bridge.set_exact_leaves(False)
        n0 = bridge.fast_leaf_passes()
        model.zero_grad(set_to_none=True)
        got = grads_after_backward()
        assert bridge.fast_leaf_passes() == n0 + 2             # both encoders took the fast path
        for a, b, (name, _) in zip(got, ref, enc):
            assert same(a, b), name
        got2 = grads_after_backward(0.5)                       # p.grad += : ONE fused add per encoder
        for a, b, (name, _) in zip(got2, ref2, enc):
            assert same(a, b), name
        # a mixed state: some leaves hold a gradient, others were reset
        for q in params[::3]:
            q.grad = None
        mixed = grads_after_backward()
        for k, (a, (name, _)) in enumerate(zip(mixed, enc)):
            want = ref[k] if k % 3 == 0 else got2[k] + ref[k]
            assert same(a, want), name
        # the functional API on a graph of its own, and on a graph that a fast backward has already walked (retain_graph)
        model.zero_grad(set_to_none=True)
        fg = torch.autograd.grad(loss_of(), params)
        for a, b, (name, _) in zip(fg, ref, enc):
            assert same(a, b), name
        loss = loss_of()
        loss.backward(retain_graph=True)
        n1 = bridge.fast_leaf_passes()
        fg2 = torch.autograd.grad(loss, params)                # the edges to the leaves are back
        assert bridge.fast_leaf_passes() == n1
        for a, b, (name, _) in zip(fg2, ref, enc):
            assert same(a, b), name
        model.zero_grad(set_to_none=True)
        loss_of().backward(inputs=params[:5])                  # backward(inputs=...): stock path, only those leaves
        assert all(q.grad is not None for q in params[:5]) and all(q.grad is None for q in params[5:])
        for a, b in zip(params[:5], ref[:5]):
            assert same(a.grad, b)
        # hooks: a tensor hook and a post-accumulate hook each see their gradient; those passes take the stock path
        model.zero_grad(set_to_none=True)
        seen = {}
        h1 = params[3].register_hook(lambda g: seen.__setitem__("pre", g.detach().clone()))
        h2 = params[-2].register_post_accumulate_grad_hook(lambda q: seen.__setitem__("post", q.grad.detach().clone()))
        n2 = bridge.fast_leaf_passes()
        hooked = grads_after_backward()
        assert bridge.fast_leaf_passes() == n2                  # both encoders had a listener: no fast pass
        assert same(seen["pre"], ref[3]) and same(seen["post"], ref[-2])
        for a, b, (name, _) in zip(hooked, ref, enc):
            assert same(a, b), name
        h1.remove(), h2.remove()
        # frozen parameters: no gradient appears there, the others are unchanged
        model.zero_grad(set_to_none=True)
        params[0].requires_grad_(False), params[-1].requires_grad_(False)
        frozen = None
        try:
            loss_of().backward()
            assert params[0].grad is None and params[-1].grad is None
            for q, b, (name, _) in list(zip(params, ref, enc))[1:-1]:
                assert same(q.grad, b), name
        finally:
            params[0].requires_grad_(True), params[-1].requires_grad_(True)
        # an optimizer step driven by the fast path moves the weights exactly as one driven by the stock path
        model.zero_grad(set_to_none=True)
    finally:
        bridge.set_exact_leaves(was)


def test_graphed_train_step_equals_eager(pretrained):
    """gvp_hip.graphed.GraphedTrainStep: the whole JointGNN training step from shape-bucketed HIP graphs.  Batches of
    different sizes (two of them share a bucket, so the second one is a pure stage + replay): loss, predictions and every
    one of the 764k gradients equal the eager step on the UNPADDED batch; p.grad holds the result."""
    from gvp_hip.graphed import GraphedTrainStep
    from models.joint_gnn import JointGNN
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
    model = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"],
                     **kw["joint_gnn_kwargs"])
    model.load_state_dict(pretrained, strict=True)
    model.to(DEV).eval()                                   # no dropout: the two runs see the same numbers
    params = [q for q in model.parameters() if q.numel()]
    from gvp_hip import _lib
    if _lib.bridge() is None:              # (CGVP_BRIDGE=0 runs of the suite) its capture-safety check lives in the bridge
        with pytest.raises(RuntimeError, match="bridge"):
            GraphedTrainStep(model, torch.nn.functional.mse_loss)
        return
    runner = GraphedTrainStep(model, torch.nn.functional.mse_loss)
    seen = set()
    for seed, lengths in ((1, [120, 80, 95, 130, 60, 101]), (2, [118, 83, 95, 128, 62, 100]), (3, [200, 150, 170, 90, 60, 210])):
        p, m = ds.pair_batch(6, seed, lengths=lengths)
        pd, md = _to(ds.to_torch(p)), _to(ds.to_torch(m))
        pd["ptr"], md["ptr"] = torch.as_tensor(p.ptr).to(DEV), torch.as_tensor(m.ptr).to(DEV)
        target = torch.randn(6, 1, device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed))
        pred, _ = model(pd, md)
        loss = torch.nn.functional.mse_loss(pred, target)
        ref = torch.autograd.grad(loss, params, retain_graph=True)
        model.zero_grad(set_to_none=True)
        if seed == 1:            # an eager graph built on ANOTHER stream is still alive: capturing now would pull that stream in
            with pytest.raises(RuntimeError, match="built on another stream"):
                runner.run(pd, md, target)
        pred, loss = pred.detach().clone(), loss.detach().clone()        # (drops the eager graph)
        gl, gp = runner.run(pd, md, target)
        torch.cuda.synchronize()
        assert float((gl - loss).abs()) <= 1e-5 * float(loss.abs())
        assert float((gp - pred).abs().max()) <= 1e-5 * float(pred.abs().max())
        scale = max(float(g.abs().max()) for g in ref)
        for q, g in zip(params, ref):
            assert q.grad is not None
            assert float((q.grad - g).abs().max()) <= 2e-5 * float(g.abs().max()) + 2e-7 * scale
        seen.add(len(runner.buckets))
    assert 1 <= len(runner.buckets) <= 3 and sum(b.steps for b in runner.buckets.values()) == 3
