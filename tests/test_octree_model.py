"""The list-free quadtree formulation (tests/octree_model.py) against the oracle's std::list restatement."""
import numpy as np
import pytest

import oracle_lib
from octree_model import octree_model


def _cands(rng, W, H, n, clustered):
    if clustered:
        cx = rng.integers(0, W - 6, 6)
        cy = rng.integers(0, H - 6, 6)
        k = rng.integers(0, 6, n * 2)
        x = np.clip(cx[k] + rng.normal(0, W / 12, n * 2), 0, W - 7).astype(int)
        y = np.clip(cy[k] + rng.normal(0, H / 12, n * 2), 0, H - 7).astype(int)
    else:
        x = rng.integers(0, W - 6, n * 2)
        y = rng.integers(0, H - 6, n * 2)
    pts = np.unique(np.stack([y, x], 1), axis=0)  # distinct pixels, (y, x) sorted
    pts = pts[rng.permutation(len(pts))[:n]]
    c = np.zeros(len(pts), oracle_lib.CAND_DTYPE)
    c["x"], c["y"] = pts[:, 1], pts[:, 0]
    c["response"] = rng.integers(7, 60, len(pts))  # many ties
    return c


@pytest.mark.parametrize("W,H", [(608, 448), (1209, 344), (720, 448), (314, 73), (147, 102), (3808, 2128)])
@pytest.mark.parametrize("clustered", [False, True])
def test_model_matches_oracle(W, H, clustered):
    rng = np.random.default_rng(W * 3 + H + clustered)
    for n, N in [(1, 5), (2, 1), (3, 3), (16, 8), (40, 100), (300, 60), (900, 217), (2500, 434), (500, 434), (435, 434)]:
        c = _cands(rng, W, H, n, clustered)
        exp = oracle_lib.distribute_octree(c, 16, 16 + W, 16, 16 + H, N)
        got = octree_model(c["x"], c["y"], c["response"], W, H, N)
        assert got == list(exp), (W, H, n, N)


def test_hand_built_small_sets():
    # two far-apart points, N=2: one sweep separates them; order = newest first
    c = np.zeros(2, oracle_lib.CAND_DTYPE)
    c["x"], c["y"], c["response"] = [10, 500], [10, 300], [9, 8]
    out = oracle_lib.distribute_octree(c, 16, 16 + 608, 16, 16 + 448, 2)
    assert list(out) == [1, 0]  # n4 (bottom-right) was pushed to the front last
    # N=1: the single root already satisfies N after the first sweep splits it (2 nodes >= 1)
    out = oracle_lib.distribute_octree(c, 16, 16 + 608, 16, 16 + 448, 1)
    assert list(out) == [1, 0]
    # three points in one quadrant cell, ties on response keep the first in candidate order
    c = np.zeros(3, oracle_lib.CAND_DTYPE)
    c["x"], c["y"], c["response"] = [1, 2, 600], [1, 2, 440], [5, 5, 5]
    out = oracle_lib.distribute_octree(c, 16, 16 + 608, 16, 16 + 448, 2)
    assert list(out) == [2, 0]
