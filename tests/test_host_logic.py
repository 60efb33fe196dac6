"""Host-side logic that needs no GPU."""
import numpy as np

from gvp_hip import ops


def test_shard_pairs_by_edges_balances_and_covers():
    rng = np.random.default_rng(0)
    edges = rng.integers(600, 12000, size=256).tolist()       # KIBA-like spread of per-pair protein edge counts
    parts = ops.shard_pairs_by_edges(edges, 8)
    assert sorted(i for p in parts for i in p) == list(range(256))
    loads = [sum(edges[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(edges)              # LPT bound: within one item of each other
    naive = [sum(edges[r * 32:(r + 1) * 32]) for r in range(8)]
    assert max(loads) <= max(naive)
    assert ops.shard_pairs_by_edges([5, 1], 4) == [[0], [1], [], []]
