"""Multi-GPU path on CPU: world_size 2, gloo.  Pairs are sharded across ranks,
`JointGNN.enable_pair_parallel()` all-gathers the per-pair embeddings before the
affinity head; the gathered result must equal the single-process result on the
whole batch, and gradients must flow through the collective.

The product's encoders are GPU-only, so THIS TEST swaps them for oracle-backed
stand-ins (test scaffolding); what is under test is the sharding + collective
logic of models/joint_gnn.py."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, PKG, REPO


class _OracleEncoder(torch.nn.Module):
    def __init__(self, fn, params, out_channels):
        super().__init__()
        self.fn, self.out_channels = fn, out_channels
        self.params = torch.nn.ParameterDict({k.replace(".", "/"): torch.nn.Parameter(v.clone())
                                              for k, v in params.items() if v.numel()})

    def forward(self, x, edge_index, ntypes, etypes, eattr=None, batch=None):
        P = {k.replace("/", "."): v for k, v in self.params.items()}
        return self.fn(P, x, edge_index, ntypes, etypes, eattr)


def _build(pretrained):
    import sys
    for p in (PKG, REPO):
        if p not in sys.path:
            sys.path.insert(0, p)
    from models.joint_gnn import JointGNN
    from oracle import gvp_oracle as O
    kw = json.load(open(os.path.join(GOLDEN, "model_kwargs.json")))
    model = JointGNN(protein_gnn_kwargs=kw["protein_gnn_kwargs"], molecule_gnn_kwargs=kw["molecule_gnn_kwargs"],
                     **kw["joint_gnn_kwargs"])
    model.load_state_dict(pretrained, strict=True)
    pp = {k[len("protein_gnn.gnn_model."):]: v for k, v in pretrained.items() if k.startswith("protein_gnn.gnn_model.")}
    mp_ = {k[len("molecule_gnn.gnn_model."):]: v for k, v in pretrained.items() if k.startswith("molecule_gnn.gnn_model.")}
    model.protein_gnn = _OracleEncoder(O.protein_lba_forward, pp, (64, 0))
    model.molecule_gnn = _OracleEncoder(O.molecule_gine_forward, mp_, 64)
    return model.eval()


def _batches(lengths, seed):
    import davis_synth as ds
    p, m = ds.pair_batch(len(lengths), seed, lengths=lengths)
    return ds.to_torch(p), ds.to_torch(m)


GRAD_KEYS = ("output_layer.weight", "pm_embed_lin.weight", "protein_lins.0.weight", "residue_lins.0.weight",
             "cross_attn_module.cross_attn_layers.0.embed1_to_2.in_proj_weight", "atom_lins.0.bias",
             "protein_gnn.params.gvp_to_scalar/ws/bias", "protein_gnn.params.conv_list/0/conv/message_func/0/ws/weight",
             "molecule_gnn.params.conv_list/0/lin/weight", "molecule_gnn.params.conv_list/1/eps")


def _grads(model):
    named = dict(model.named_parameters())
    return {k: named[k].grad.detach().clone().numpy() for k in GRAD_KEYS}


def _worker(rank, world, port, pretrained, shards, pass_counts, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        model = _build(pretrained).enable_pair_parallel(
            pair_counts=[len(s) for s in shards] if pass_counts else None)
        pd, md = _batches(shards[rank], 100 + rank)
        y, _ = model(pd, md)
        y.square().sum().backward()
        local = _grads(model)                       # before the reduce: encoder gradients are rank-local
        model.reduce_pair_parallel_grads()
        q.put((rank, y.detach().numpy(), local, _grads(model)))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("shards,pass_counts", [([[40, 33, 52, 28], [47, 30]], True),      # ragged: 4 + 2 pairs
                                                ([[40, 33, 52], [47, 30, 28]], False)])    # equal shards, no counts
def test_pair_parallel_world2_matches_single_process(pretrained, shards, pass_counts):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, pretrained, shards, pass_counts, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r = q.get(timeout=240)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    # single process: the same six pairs; the loss sum(y^2) is separable over pairs, so the whole-batch
    # gradient is the sum of the per-shard backward passes
    model = _build(pretrained)
    outs = []
    for rank, lens in enumerate(shards):
        pd, md = _batches(lens, 100 + rank)
        y = model(pd, md)[0]
        y.square().sum().backward()
        outs.append(y.detach().numpy())
    full, ref = np.concatenate(outs), _grads(model)
    for rank in (0, 1):
        y = res[rank][0]
        assert y.shape == (6, 1)                         # every rank holds ALL predictions
        assert np.allclose(y, full, rtol=1e-5, atol=1e-6)
    assert np.allclose(res[0][0], res[1][0])
    for k in GRAD_KEYS:
        scale = np.abs(ref[k]).max()
        assert scale > 0, k
        for rank in (0, 1):                              # after the reduce: every rank holds the 6-pair gradient
            assert np.abs(res[rank][2][k] - ref[k]).max() <= 2e-4 * scale + 1e-7, (k, rank)
    # before the reduce: head gradients are already complete, encoder gradients are rank-local (and differ)
    for k in ("output_layer.weight", "pm_embed_lin.weight"):
        assert np.abs(res[0][1][k] - ref[k]).max() <= 2e-4 * np.abs(ref[k]).max() + 1e-7
    k = "protein_gnn.params.gvp_to_scalar/ws/bias"
    assert np.abs(res[0][1][k] - ref[k]).max() > 1e-3 * np.abs(ref[k]).max()
    assert np.abs(res[0][1][k] + res[1][1][k] - ref[k]).max() <= 2e-4 * np.abs(ref[k]).max() + 1e-7


@pytest.mark.timeout(420)
def test_pair_parallel_world4_with_edge_balanced_ragged_shards(pretrained):
    """world_size 4 over gloo, shards chosen by `shard_pairs_by_edges` (the reference's edge-balanced sampler,
    dataset/dual_dataset.py:476-516, turned into per-rank sharding): ranks hold 2-4 pairs each; the gathered predictions
    and every reduced gradient equal the single-process run over the same 11 pairs."""
    import sys
    for p in (PKG, REPO):
        if p not in sys.path:
            sys.path.insert(0, p)
    from gvp_hip.ops import shard_pairs_by_edges
    lengths = [61, 24, 35, 48, 19, 55, 30, 27, 44, 22, 38]
    edge_counts = [3 * n for n in lengths]                       # ~3 edges per residue at 4 A: what the sampler balances
    parts = shard_pairs_by_edges(edge_counts, 4)
    assert sorted(i for p in parts for i in p) == list(range(len(lengths))) and all(parts)
    assert len({len(p) for p in parts}) > 1                      # ragged
    loads = [sum(edge_counts[i] for i in p) for p in parts]
    assert max(loads) <= sum(edge_counts) / 4 + max(edge_counts)
    shards = [[lengths[i] for i in p] for p in parts]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 4, port, pretrained, shards, True, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r = q.get(timeout=360)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    model = _build(pretrained)
    outs = []
    for rank, lens in enumerate(shards):
        pd, md = _batches(lens, 100 + rank)
        y = model(pd, md)[0]
        y.square().sum().backward()
        outs.append(y.detach().numpy())
    full, ref = np.concatenate(outs), _grads(model)
    for rank in range(4):
        assert res[rank][0].shape == (len(lengths), 1)
        assert np.allclose(res[rank][0], full, rtol=1e-5, atol=1e-6)
        for k in GRAD_KEYS:
            scale = np.abs(ref[k]).max()
            assert np.abs(res[rank][2][k] - ref[k]).max() <= 2e-4 * scale + 1e-7, (k, rank)
