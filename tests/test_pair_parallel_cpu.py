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


def _worker(rank, world, port, pretrained, shards, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        model = _build(pretrained).enable_pair_parallel()
        pd, md = _batches(shards[rank], 100 + rank)
        y, _ = model(pd, md)
        y.square().sum().backward()
        g = model.output_layer.weight.grad.clone()
        ge = model.protein_gnn.params["gvp_to_scalar/ws/bias"].grad.clone()
        q.put((rank, y.detach().numpy(), g.numpy(), ge.numpy()))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_pair_parallel_world2_matches_single_process(pretrained):
    shards = [[40, 33, 52, 28], [47, 30]]                 # ragged: 4 + 2 pairs
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, pretrained, shards, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r = q.get(timeout=240)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    # single process: the same six pairs through the head in one go
    model = _build(pretrained)
    outs = []
    for rank, lens in enumerate(shards):
        pd, md = _batches(lens, 100 + rank)
        outs.append(model(pd, md)[0].detach().numpy())
    full = np.concatenate(outs)                          # pairs are independent: per-shard == whole batch
    for rank in (0, 1):
        y = res[rank][0]
        assert y.shape == (6, 1)                         # every rank holds ALL predictions
        assert np.allclose(y, full, rtol=1e-5, atol=1e-6)
    assert np.allclose(res[0][0], res[1][0])
    # the head is replicated: identical head gradients; encoder gradients are local and non-zero
    assert np.allclose(res[0][1], res[1][1], rtol=1e-5, atol=1e-7)
    assert np.abs(res[0][2]).max() > 0 and np.abs(res[1][2]).max() > 0
