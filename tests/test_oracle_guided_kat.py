"""Known-answer tests of the oracle's restatement of the matching loops of ORBmatcher::SearchByProjection(CurrentFrame, pKF,
sAlreadyFound, th, ORBdist) (reference src/ORBmatcher.cc:1565-1632) and SearchByProjection(pKF, Scw, vpPoints, vpMatched,
th) (:362-398) on projected map points."""
import numpy as np

import oracle_lib

GRID = (0.0, 0.0, 64 / 640.0, 48 / 480.0)


def ones(n):
    return np.packbits(np.arange(256) < n)


def frame(oracle, feats):
    """feats: (x, y, octave, bits[, angle]) per feature."""
    xy = np.array([(p[0], p[1]) for p in feats], np.float32).reshape(-1, 2)
    start, idx = oracle.assign_grid(xy, *GRID)
    return dict(xy=xy, octave=np.array([p[2] for p in feats], np.int32), desc=np.stack([ones(p[3]) for p in feats]),
                angle=np.array([p[4] if len(p) > 4 else 0.0 for p in feats], np.float32), start=start, idx=idx)


def point(u, v, level=0, radius=5.0, bits=0, angle=0.0, flags=1):
    q = np.zeros((), oracle_lib.GUIDED_QUERY_DTYPE)
    q["u"], q["v"], q["angle"], q["level"], q["radius"], q["flags"] = u, v, angle, level, radius, flags
    q["desc"][:] = ones(bits)
    return q


def run(F, points, level_up=1, max_dist=100, check=False, taken0=None):
    q = np.array(points, oracle_lib.GUIDED_QUERY_DTYPE)
    return oracle_lib.search_guided(F["xy"], F["octave"], F["angle"], F["desc"], taken0, F["start"], F["idx"], GRID[0], GRID[1],
                                    GRID[2], GRID[3], q, level_up, max_dist, check)


def test_level_ranges_of_the_two_functions(oracle):
    F = frame(oracle, [(100, 100, 1, 30), (101, 100, 2, 20), (102, 100, 3, 10), (103, 100, 4, 0)])
    # relocalisation: levels predicted - 1 .. predicted + 1 (:1565): octaves 1, 2, 3 qualify at level 2 -> distance 10
    n, m, qm = run(F, [point(101, 100, level=2)], level_up=1)
    assert n == 1 and qm[0] == 2
    # loop closing: predicted - 1 .. predicted (:380): octaves 1 and 2 -> distance 20
    n, m, qm = run(F, [point(101, 100, level=2)], level_up=0)
    assert n == 1 and qm[0] == 1


def test_thresholds_and_skipped_points(oracle):
    F = frame(oracle, [(100, 100, 0, 64)])
    assert run(F, [point(100, 100)], max_dist=64)[0] == 1        # bestDist <= ORBdist (:1592)
    assert run(F, [point(100, 100)], max_dist=63)[0] == 0
    assert run(F, [point(100, 100, flags=3)], max_dist=100)[0] == 0   # a point the reference skips
    # the window is open and centred on the projection
    assert run(F, [point(105, 100, radius=5.0)])[0] == 0
    assert run(F, [point(104.9, 100, radius=5.0)])[0] == 1


def test_features_with_a_map_point_are_passed_over(oracle):
    # feature 0 holds a map point before the call (:1578, :375): the point takes the farther feature 1
    F = frame(oracle, [(100, 100, 0, 0), (101, 100, 0, 30)])
    n, m, qm = run(F, [point(100, 100)], taken0=np.array([1, 0], np.uint8))
    assert n == 1 and qm[0] == 1 and list(m) == [-1, 0]
    # two points want feature 0: the first takes it (:1594, :396), the second sees only feature 1
    n, m, qm = run(F, [point(100, 100, bits=5), point(100, 100)])
    assert n == 2 and list(qm) == [0, 1] and list(m) == [0, 1]
    # ... and finds nothing when feature 1 is too far for the threshold: a better later claim does not take over
    n, m, qm = run(F, [point(100, 100, bits=5), point(100, 100)], max_dist=20)
    assert n == 1 and list(qm) == [0, -1]


def test_rotation_histogram(oracle):
    # eleven matches rotate by 0 degrees, one by 90: 1 < 0.1 * 11 -> its feature is released (:1614-1633)
    feats = [(50 + 40 * i, 100, 0, 0, 0.0) for i in range(12)]
    F = frame(oracle, feats)
    pts = [point(50 + 40 * i, 100, angle=0.0 if i < 11 else 90.0) for i in range(12)]
    n, m, qm = run(F, pts, check=True)
    assert n == 11 and qm[11] == -1 and m[11] == -1 and list(qm[:11]) == list(range(11))
    assert run(F, pts, check=False)[0] == 12
