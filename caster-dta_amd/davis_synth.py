"""Seeded synthetic "Davis-shaped" protein / drug graph batches.

No structures ship with the reference (only sequences + SMILES), so every
test, golden vector and bench line runs on graphs generated here with the
shapes, value ranges and edge ordering the reference's featurisers produce:

* residue graph   -- utils/create_protein_features.py:34-109 (node features),
                     :225-273 (RBF / positional / direction edge features),
                     :290-331 (``dist`` and ``num`` edge thresholds),
                     utils/create_graphs.py:29,44-52 (row-major COO over
                     (src, dst), i.e. ``edge_index[0]`` non-decreasing)
* drug graph      -- utils/create_smiles_features.py:23-104, utils/smiles_utils.py:51-61
* batching        -- dataset/dual_dataset.py:538-547 (PyG ``Batch.from_data_list``:
                     concatenate with node offsets, ``batch`` vector, ``ptr``)

Everything is numpy on the host; ``to_torch`` moves a batch to a device.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

CA_STEP = 3.8          # Angstrom between consecutive C-alpha atoms
RBF_COUNT = 16
RBF_MAX = 20.0
POS_EMBEDS = 16


@dataclass
class GraphBatch:
    """A PyG-style batched graph: node/edge tables plus ptr/batch offsets."""
    x_s: np.ndarray                 # [N, Fs] float32
    x_v: np.ndarray | None          # [N, Fv, 3] float32 (protein only)
    edge_index: np.ndarray          # [2, E] int64, row 0 = source, row 1 = target
    e_s: np.ndarray                 # [E, Es] float32
    e_v: np.ndarray | None          # [E, Ev, 3] float32 (protein only)
    ntypes: np.ndarray              # [N] int64
    etypes: np.ndarray              # [E] int64
    batch: np.ndarray               # [N] int64 graph id per node
    ptr: np.ndarray                 # [B+1] int64 node offsets
    eptr: np.ndarray = field(default=None)  # [B+1] int64 edge offsets

    @property
    def num_nodes(self):
        return int(self.x_s.shape[0])

    @property
    def num_edges(self):
        return int(self.edge_index.shape[1])

    @property
    def num_graphs(self):
        return int(self.ptr.shape[0] - 1)


def _unit(v, axis=-1):
    n = np.linalg.norm(v, axis=axis, keepdims=True)
    return np.divide(v, n, out=np.zeros_like(v), where=n != 0)


def ca_trace(length, rng):
    """C-alpha-like chain: fixed 3.8 A virtual bonds, bond angle 85..145 deg,
    free dihedral, so |CA_i - CA_{i+2}| is 5.1..7.2 A as in real proteins."""
    pts = np.zeros((length, 3))
    if length > 1:
        pts[1] = [CA_STEP, 0.0, 0.0]
    for i in range(2, length):
        b = _unit(pts[i - 1] - pts[i - 2])
        ref = np.array([0.0, 0.0, 1.0]) if abs(b[2]) < 0.9 else np.array([1.0, 0.0, 0.0])
        if i >= 3:
            prev = pts[i - 2] - pts[i - 3]
            nrm = np.cross(prev, b)
            if np.linalg.norm(nrm) > 1e-6:
                ref = nrm
        n1 = _unit(np.cross(b, ref))
        n2 = np.cross(b, n1)
        for _ in range(32):                                # self-avoiding: no non-bonded CA within 4.2 A
            theta = np.deg2rad(rng.uniform(85.0, 145.0))  # virtual bond angle
            phi = rng.uniform(-np.pi, np.pi)               # dihedral
            d = -np.cos(theta) * b + np.sin(theta) * (np.cos(phi) * n1 + np.sin(phi) * n2)
            cand = pts[i - 1] + CA_STEP * d
            if i < 3 or np.min(np.linalg.norm(pts[:i - 1] - cand, axis=-1)) > 4.2:
                break
        pts[i] = cand
    return pts


def _pos_encoding(idx_diff):
    per = POS_EMBEDS // 2
    freqs = np.exp(2 * np.arange(per) * -(np.log(10000.0) / per))
    ang = idx_diff[..., None] * freqs
    return np.concatenate([np.cos(ang), np.sin(ang)], axis=-1)


def protein_graph(length, rng, thresh=4.0, thresh_type="dist", num_ntypes=20):
    """One residue graph with node (17,3) and edge (32,1) features."""
    ca = ca_trace(length, rng)
    diff = ca[:, None, :] - ca[None, :, :]               # [i, j] = CA_i - CA_j
    dist = np.linalg.norm(diff, axis=-1)
    if thresh_type == "dist":
        keep = dist <= thresh                             # includes self loops
    elif thresh_type == "num":
        k = int(min(thresh, length))
        nearest = np.argsort(dist, axis=-1, kind="stable")[:, :k]
        keep = np.zeros_like(dist, dtype=bool)
        keep[np.arange(length)[:, None], nearest] = True
    else:
        raise ValueError(thresh_type)
    src, dst = np.nonzero(keep)                           # row-major => src sorted
    d = dist[src, dst]
    mu = np.linspace(0.0, RBF_MAX, RBF_COUNT)
    step = RBF_MAX / RBF_COUNT
    rbf = np.exp(-np.square((d[:, None] - mu[None, :]) / step))
    pos = _pos_encoding((dst - src).astype(np.float64))
    e_s = np.concatenate([rbf, pos], axis=-1).astype(np.float32)
    e_v = _unit(diff[src, dst])[:, None, :].astype(np.float32)   # zero on self loops

    ang = rng.uniform(-np.pi, np.pi, size=(length, 3))
    dihedral = np.concatenate([np.cos(ang), np.sin(ang)], axis=-1)
    dihedral[0, [0, 3]] = 0.0                              # undefined phi at N-terminus
    dihedral[-1, [1, 2, 4, 5]] = 0.0                       # undefined psi/omega at C-terminus
    props = rng.uniform(0.0, 1.0, size=(length, 11))
    x_s = np.concatenate([dihedral, props], axis=-1).astype(np.float32)
    fwd = np.zeros((length, 3))
    bwd = np.zeros((length, 3))
    fwd[:-1] = _unit(ca[1:] - ca[:-1])
    bwd[1:] = _unit(ca[:-1] - ca[1:])
    side = _unit(rng.normal(size=(length, 3)))
    x_v = np.stack([fwd, bwd, side], axis=1).astype(np.float32)
    ntypes = rng.integers(0, num_ntypes, size=length).astype(np.int64)
    etypes = np.zeros(src.shape[0], dtype=np.int64)
    ei = np.stack([src, dst]).astype(np.int64)
    return dict(x_s=x_s, x_v=x_v, edge_index=ei, e_s=e_s, e_v=e_v, ntypes=ntypes, etypes=etypes)


_ATOM_GROUPS = (3, 7, 6, 12, 8)   # one-hot blocks: 36 of the 41 atom features


def drug_graph(rng, n_atoms=None, num_ntypes=11):
    """One molecular graph: random tree + ~10% ring closures, both bond
    directions plus self loops; 41 atom features, 9 bond features."""
    if n_atoms is None:
        n_atoms = int(np.clip(np.rint(rng.normal(40.0, 8.0)), 10, 100))
    bonds = set()
    for a in range(1, n_atoms):
        p = int(rng.integers(max(0, a - 4), a))
        bonds.add((p, a))
    n_ring = max(0, int(round(0.1 * n_atoms)))
    for _ in range(n_ring):
        a = int(rng.integers(0, n_atoms))
        b = int(np.clip(a + rng.integers(3, 7), 0, n_atoms - 1))
        if a != b and (min(a, b), max(a, b)) not in bonds:
            bonds.add((min(a, b), max(a, b)))
    adj = np.zeros((n_atoms, n_atoms), dtype=np.int64)    # 0 = no edge, else bond class
    battr = {}
    for (a, b) in bonds:
        cls = int(rng.integers(1, 5))
        feat = np.zeros(9, dtype=np.float32)
        feat[int(rng.integers(0, 7))] = 1.0
        feat[7] = float(rng.integers(0, 2))
        feat[8] = float(rng.integers(0, 2))
        adj[a, b] = adj[b, a] = cls
        battr[(a, b)] = battr[(b, a)] = feat
    src, dst = [], []
    etypes, eattr = [], []
    for i in range(n_atoms):
        for j in range(n_atoms):
            if i == j:
                src.append(i); dst.append(j); etypes.append(0)
                eattr.append(np.zeros(9, dtype=np.float32))
            elif adj[i, j]:
                src.append(i); dst.append(j); etypes.append(int(adj[i, j]))
                eattr.append(battr[(i, j)])
    x = np.zeros((n_atoms, 41), dtype=np.float32)
    col = 0
    for g in _ATOM_GROUPS:
        x[np.arange(n_atoms), col + rng.integers(0, g, size=n_atoms)] = 1.0
        col += g
    x[:, 36] = rng.integers(-1, 2, size=n_atoms)           # formal charge
    x[:, 37] = 0.0                                          # radical electrons
    x[:, 38] = rng.integers(0, 2, size=n_atoms)            # aromatic
    x[:, 39] = rng.integers(0, 2, size=n_atoms)            # in ring
    x[:, 40] = rng.normal(0.0, 0.2, size=n_atoms)          # Gasteiger charge
    ntypes = rng.integers(0, num_ntypes, size=n_atoms).astype(np.int64)
    return dict(x_s=x, x_v=None, edge_index=np.array([src, dst], dtype=np.int64),
                e_s=np.stack(eattr).astype(np.float32), e_v=None, ntypes=ntypes,
                etypes=np.array(etypes, dtype=np.int64))


def collate(graphs):
    """Concatenate graphs with node offsets (dataset/dual_dataset.py:543-544)."""
    n_off, e_off = 0, 0
    ptr, eptr = [0], [0]
    xs, xv, ei, es, ev, nt, et, bt = [], [], [], [], [], [], [], []
    for gi, g in enumerate(graphs):
        n = g["x_s"].shape[0]
        xs.append(g["x_s"]); nt.append(g["ntypes"]); et.append(g["etypes"])
        es.append(g["e_s"]); ei.append(g["edge_index"] + n_off)
        if g["x_v"] is not None:
            xv.append(g["x_v"]); ev.append(g["e_v"])
        bt.append(np.full(n, gi, dtype=np.int64))
        n_off += n; e_off += g["edge_index"].shape[1]
        ptr.append(n_off); eptr.append(e_off)
    vec = len(xv) > 0
    return GraphBatch(
        x_s=np.concatenate(xs), x_v=np.concatenate(xv) if vec else None,
        edge_index=np.concatenate(ei, axis=1), e_s=np.concatenate(es),
        e_v=np.concatenate(ev) if vec else None, ntypes=np.concatenate(nt),
        etypes=np.concatenate(et), batch=np.concatenate(bt),
        ptr=np.array(ptr, dtype=np.int64), eptr=np.array(eptr, dtype=np.int64))


def protein_batch(n_graphs, seed, length=300, thresh=4.0, thresh_type="dist",
                  lengths=None, num_ntypes=20):
    rng = np.random.default_rng(seed)
    if lengths is None:
        lengths = [length] * n_graphs
    return collate([protein_graph(int(L), rng, thresh, thresh_type, num_ntypes) for L in lengths])


def drug_batch(n_graphs, seed, n_atoms=None, num_ntypes=11):
    rng = np.random.default_rng(seed + 7919)
    return collate([drug_graph(rng, n_atoms, num_ntypes) for _ in range(n_graphs)])


def real_lengths(dataset, n, seed):
    """`n` protein lengths drawn (seeded, with replacement) from the sequence lengths of the reference's
    Davis / KIBA protein tables (data/deepdta_data/{davis,kiba}/proteins.txt; only the integer lengths are
    kept, in tests/golden/protein_lengths.json -- generator: tests/golden/make_length_stats.py)."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                        "protein_lengths.json")
    if dataset == "bindingdb":
        # no BindingDB sequences ship with the reference; its length filter is 25..3000 (dataset/load_data.py:218-222)
        # and the shipped example batch averages 558 residues (model_summary.txt): the KIBA table rescaled to that mean
        table = np.clip(np.rint(np.asarray(json.load(open(path))["kiba"]) * (558.0 / 729.0)), 25, 3000).astype(int)
    else:
        table = np.asarray(json.load(open(path))[dataset])
    return [int(v) for v in np.random.default_rng(seed).choice(table, size=n, replace=True)]


def pair_batch(n_pairs, seed, length=300, thresh=4.0, thresh_type="dist", lengths=None):
    """One batch of protein/drug pairs (the unit of the graph-pairs/sec metric)."""
    return (protein_batch(n_pairs, seed, length, thresh, thresh_type, lengths),
            drug_batch(n_pairs, seed))


def to_torch(gb, device="cpu", dtype=None):
    """GraphBatch -> the kwargs dict JointGNN._graphs_to_dicts builds
    (models/joint_gnn.py:151-170): x, edge_index, ntypes, etypes, eattr, batch."""
    import torch
    dtype = dtype or torch.float32
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)
    i = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device=device)
    if gb.x_v is not None:
        x = (f(gb.x_s), f(gb.x_v)); eattr = (f(gb.e_s), f(gb.e_v))
    else:
        x = f(gb.x_s); eattr = f(gb.e_s)
    return dict(x=x, edge_index=i(gb.edge_index), ntypes=i(gb.ntypes), etypes=i(gb.etypes),
                eattr=eattr, batch=i(gb.batch))
