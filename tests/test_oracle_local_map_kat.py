"""Known-answer tests of the oracle's ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)
(reference src/ORBmatcher.cc:45-129): hand-built frames whose result can be read off the source."""
import numpy as np

import oracle_lib

GRID = dict(min_x=0.0, min_y=0.0, inv_w=64 / 640.0, inv_h=48 / 480.0)
SCALE = np.array([1.2 ** i for i in range(8)], np.float32)


def frame(oracle, pts, octaves=None):
    xy = np.array(pts, np.float32)
    start, idx = oracle.assign_grid(xy, 0.0, 0.0, GRID["inv_w"], GRID["inv_h"])
    n = len(pts)
    return dict(xy=xy, octave=np.zeros(n, np.int32) if octaves is None else np.array(octaves, np.int32),
                desc=np.zeros((n, 32), np.uint8), start=start, idx=idx)


def point(u, v, level=0, view_cos=1.0, flags=1, ur=0.0, bits=0):
    q = np.zeros((), oracle_lib.LOCAL_QUERY_DTYPE)
    q["u"], q["v"], q["ur"], q["level"], q["view_cos"], q["flags"] = u, v, ur, level, view_cos, flags
    q["desc"][:] = np.packbits(np.arange(256) < bits)      # `bits` leading ones: distance to a zero descriptor
    return q


def run(F, queries, th=1.0, ratio=0.8, u_right=None, taken0=None):
    q = np.array(queries, oracle_lib.LOCAL_QUERY_DTYPE)
    return oracle_lib.search_local_map(F["xy"], F["octave"], F["desc"], u_right, taken0, F["start"], F["idx"], SCALE,
                                       GRID["min_x"], GRID["min_y"], GRID["inv_w"], GRID["inv_h"], th, ratio, q)


def ones(n):
    return np.packbits(np.arange(256) < n)


def test_radius_by_viewing_cos_and_th(oracle):
    F = frame(oracle, [(103, 100)])
    # viewCos > 0.998 -> r = 2.5: a feature 3 px away is outside; else r = 4.0: inside (:131-137)
    assert run(F, [point(100, 100, view_cos=0.999)])[0] == 0
    assert run(F, [point(100, 100, view_cos=0.99)])[0] == 1
    # th != 1 multiplies the radius (:65-66); the window scales with the predicted level (:69)
    assert run(F, [point(100, 100, view_cos=0.999)], th=3.0)[0] == 1
    F1 = frame(oracle, [(102.9, 100)], octaves=[1])
    assert run(F1, [point(100, 100, level=1, view_cos=0.999)])[0] == 1       # 2.5 * 1.2 = 3.0 > 2.9
    # the window is open: |dx| < r strictly (src/Frame.cc:389)
    F2 = frame(oracle, [(104, 100)])
    assert run(F2, [point(100, 100, view_cos=0.5)])[0] == 0


def test_level_range_is_predicted_minus_one_to_predicted(oracle):
    F = frame(oracle, [(100, 100), (101, 100), (102, 100)], octaves=[1, 2, 3])
    n, match, qm = run(F, [point(100, 100, level=2, view_cos=0.5)])
    assert n == 1 and qm[0] == 0          # octaves 1 and 2 qualify, equal distances: the first visited wins
    n, match, qm = run(F, [point(100, 100, level=3, view_cos=0.5)])
    assert qm[0] == 1                      # octaves 2 and 3
    F0 = frame(oracle, [(100, 100)], octaves=[0])
    assert run(F0, [point(100, 100, level=0, view_cos=0.5)])[0] == 1          # minLevel -1 excludes nothing


def test_ratio_test_only_within_one_level(oracle):
    # best 40, second 45 on the same level: 40 > 0.8 * 45 = 36 -> rejected (:120-121)
    F = frame(oracle, [(100, 100), (101, 100)])
    F["desc"][0], F["desc"][1] = ones(40), ones(45)
    assert run(F, [point(100, 100, view_cos=0.5)])[0] == 0
    # second far enough: 40 <= 0.8 * 51 = 40.8 -> accepted
    F["desc"][1] = ones(51)
    n, match, qm = run(F, [point(100, 100, view_cos=0.5)])
    assert n == 1 and qm[0] == 0 and list(match) == [0, -1]
    # second on another level: no ratio test
    F = frame(oracle, [(100, 100), (101, 100)], octaves=[1, 0])
    F["desc"][0], F["desc"][1] = ones(40), ones(45)
    assert run(F, [point(100, 100, level=1, view_cos=0.5)])[0] == 1
    # TH_HIGH = 100 inclusive (:118)
    F = frame(oracle, [(100, 100)])
    F["desc"][0] = ones(100)
    assert run(F, [point(100, 100, view_cos=0.5)])[0] == 1
    F["desc"][0] = ones(101)
    assert run(F, [point(100, 100, view_cos=0.5)])[0] == 0


def test_second_best_follows_the_sequential_update(oracle):
    # visiting order a(50) b(50) c(30): c becomes best and demotes a; 30 > 0.8 * 50 is false -> accepted
    F = frame(oracle, [(100, 100), (100.5, 100), (101, 100)])
    F["desc"][0], F["desc"][1], F["desc"][2] = ones(50), ones(50), ones(30)
    n, match, qm = run(F, [point(100, 100, view_cos=0.5)])
    assert n == 1 and qm[0] == 2
    # equal best and second on one level: d > 0.8 d -> rejected
    F["desc"][2] = ones(50)
    assert run(F, [point(100, 100, view_cos=0.5)])[0] == 0


def test_held_features_and_overwrites(oracle):
    F = frame(oracle, [(100, 100), (102, 100)])
    F["desc"][1] = ones(60)
    # both points prefer feature 0; the first (with observations) holds it, the second then sees only
    # feature 1 (no second best: level -1) and takes it
    n, match, qm = run(F, [point(100, 100, view_cos=0.5), point(100, 100, view_cos=0.5)])
    assert n == 2 and list(qm) == [0, 1] and list(match) == [0, 1]
    # a first point without observations does not hold it: the second overwrites, both count (:123-124);
    # for the second point feature 1 is the second best on the same level: 0 > 0.8 * 60 is false
    n, match, qm = run(F, [point(100, 100, view_cos=0.5, flags=0), point(100, 100, view_cos=0.5)])
    assert n == 2 and list(qm) == [0, 0] and list(match) == [1, -1]
    # held before the call (:87-89); skipped points (:54-58)
    n, match, qm = run(F, [point(100, 100, view_cos=0.5, flags=3), point(100, 100, view_cos=0.5)],
                       taken0=np.array([1, 0], np.uint8))
    assert n == 1 and list(qm) == [-1, 1] and list(match) == [-1, 1]
    # taking away a point's second best changes its ratio test: alone, point B (best 40 at feature 0, second 45 at
    # feature 1) is rejected; after point A has taken feature 1, B's second is gone and B is accepted
    F = frame(oracle, [(100, 100), (101, 100)])
    F["desc"][0], F["desc"][1] = ones(40), ones(45)
    A = point(101, 100, view_cos=0.999, bits=45)          # r = 2.5 around (101, 100): both features, feature 1 at distance 0
    B = point(100, 100, view_cos=0.5)
    assert run(F, [B])[0] == 0
    n, match, qm = run(F, [A, B])
    assert n == 2 and list(qm) == [1, 0]


def test_right_coordinate_gate(oracle):
    F = frame(oracle, [(100, 100), (101, 100)])
    F["desc"][1] = ones(70)
    ur = np.array([50.0, -1.0], np.float32)
    # feature 0 has a right coordinate 6 px from the predicted one: outside r = 4 (:91-96); feature 1 has none
    n, match, qm = run(F, [point(100, 100, view_cos=0.5, ur=56.0)], u_right=ur)
    assert n == 1 and qm[0] == 1
    n, match, qm = run(F, [point(100, 100, view_cos=0.5, ur=53.9)], u_right=ur)
    assert qm[0] == 0
