"""Known-answer tests of the oracle's restatement of the search inside ORBmatcher::Fuse(pKF, vpMapPoints, th)
(reference src/ORBmatcher.cc:929-986): hand-built keyframes whose result can be read off the source."""
import numpy as np

import oracle_lib

GRID = (0.0, 0.0, 64 / 640.0, 48 / 480.0)
SCALE = np.array([1.2 ** i for i in range(8)], np.float32)
INV_SIGMA2 = (np.float32(1.0) / (SCALE * SCALE)).astype(np.float32)


def ones(n):
    return np.packbits(np.arange(256) < n)


def run(oracle, feats, u, v, radius, level, ur=0.0, bits=0):
    """feats: (x, y, octave, bits[, u_right]) per keypoint of the keyframe; the map point's descriptor has `bits` ones."""
    xy = np.array([(p[0], p[1]) for p in feats], np.float32).reshape(-1, 2)
    octave = np.array([p[2] for p in feats], np.int32)
    desc = np.stack([ones(p[3]) for p in feats])
    u_right = np.array([p[4] if len(p) > 4 else -1.0 for p in feats], np.float32)
    start, idx = oracle.assign_grid(xy, *GRID)
    return oracle_lib.fuse_search(xy, octave, desc, u_right, start, idx, GRID[0], GRID[1], GRID[2], GRID[3], u, v, ur, radius, level,
                                  INV_SIGMA2, ones(bits))


def test_levels_window_and_best_distance(oracle):
    # predicted level 2: octaves 1 and 2 qualify, 0 and 3 do not (:948)
    feats = [(100, 100, 0, 0), (101, 100, 3, 0), (100, 101, 1, 20), (101, 101, 2, 10)]
    assert run(oracle, feats, 100, 100, 5.0, 2) == (3, 10)
    # predicted level 0: only octave 0
    assert run(oracle, feats, 100, 100, 5.0, 0) == (0, 0)
    # equal distances: the first in grid order wins (dist < bestDist, :981)
    feats = [(100, 100, 0, 10), (100.5, 100, 0, 10)]
    assert run(oracle, feats, 100, 100, 5.0, 0) == (0, 10)
    # nothing in the window: -1 and 256
    assert run(oracle, [(300, 300, 0, 0)], 100, 100, 5.0, 0) == (-1, 256)


def test_chi_square_gates(oracle):
    # monocular: (ex^2 + ey^2) / sigma2 > 5.99 rejects (:965-975).  Level 0: 2.44 px passes (5.95), 2.45 px does not (6.0025)
    assert run(oracle, [(102.44, 100, 0, 0)], 100, 100, 5.0, 0)[0] == 0
    assert run(oracle, [(102.45, 100, 0, 0)], 100, 100, 5.0, 0)[0] == -1
    # level 1 (sigma2 1.44): 2.93 px passes (8.5849 / 1.44 = 5.96), 2.95 px does not (6.04)
    assert run(oracle, [(102.93, 100, 1, 0)], 100, 100, 5.0, 1)[0] == 0
    assert run(oracle, [(102.95, 100, 1, 0)], 100, 100, 5.0, 1)[0] == -1
    # with a stereo measurement the right coordinate counts and the limit is 7.8 (:951-964): ex = 2, er = 1.9 -> 7.61
    # passes, er = 2 -> 8 does not; a keypoint with mvuRight == 0 is a stereo keypoint too (>= 0)
    assert run(oracle, [(102, 100, 0, 0, 61.9)], 100, 100, 5.0, 0, ur=60.0)[0] == 0
    assert run(oracle, [(102, 100, 0, 0, 62.0)], 100, 100, 5.0, 0, ur=60.0)[0] == -1
    assert run(oracle, [(102, 100, 0, 0, 0.0)], 100, 100, 5.0, 0, ur=1.5)[0] == 0      # 4 + 2.25 = 6.25 <= 7.8
    assert run(oracle, [(102.6, 100, 0, 0, 0.0)], 100, 100, 5.0, 0, ur=0.0)[0] == 0    # 6.76 <= 7.8, above the monocular limit
    # a closer descriptor outside the gate loses to a farther one inside
    assert run(oracle, [(102.6, 100, 0, 0), (101, 100, 0, 30)], 100, 100, 5.0, 0) == (1, 30)
