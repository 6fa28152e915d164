"""Known-answer tests of the oracle's ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:418-549):
hand-built frame pairs whose result can be read off the source."""
import numpy as np

import oracle_lib

GRID = (0.0, 0.0, 64 / 640.0, 48 / 480.0)


def ones(n):
    """A descriptor with n leading one bits: distance n to the zero descriptor."""
    return np.packbits(np.arange(256) < n)


def run(oracle, f1, f2, prev=None, window=100, ratio=0.9, check=False):
    """f1: list of (x, y, octave, bits, angle) of the first frame, f2 likewise of the second; the descriptors are
    `bits` leading ones (an int) or a ready 32-byte array.  prev: vbPrevMatched, default the first frame's positions."""
    def side(feats):
        xy = np.array([(p[0], p[1]) for p in feats], np.float32).reshape(-1, 2)
        octave = np.array([p[2] for p in feats], np.int32)
        desc = np.stack([ones(p[3]) if np.isscalar(p[3]) else p[3] for p in feats]) if feats else np.zeros((0, 32), np.uint8)
        angle = np.array([p[4] if len(p) > 4 else 0.0 for p in feats], np.float32)
        return xy, octave, desc, angle
    xy1, o1, d1, a1 = side(f1)
    xy2, o2, d2, a2 = side(f2)
    start, idx = oracle.assign_grid(xy2, *GRID)
    pm = xy1 if prev is None else np.array(prev, np.float32)
    return oracle_lib.search_for_initialization(o1, a1, d1, pm, xy2, o2, a2, d2, start, idx, GRID[0], GRID[1], GRID[2], GRID[3],
                                                window, ratio, check)


def test_level_zero_only_and_the_window(oracle):
    # a keypoint of level 1 does not search at all (:438); features of the second frame above level 0 are no
    # candidates (GetFeaturesInArea with minLevel = maxLevel = 0, :444)
    n, m, pm = run(oracle, [(100, 100, 1, 0)], [(100, 100, 0, 0)])
    assert n == 0 and list(m) == [-1]
    n, m, pm = run(oracle, [(100, 100, 0, 0)], [(100, 100, 1, 0), (120, 100, 0, 10)])
    assert n == 1 and list(m) == [1]
    # the window is open: |dx| < windowSize strictly (src/Frame.cc:389)
    n, m, pm = run(oracle, [(100, 100, 0, 0)], [(200, 100, 0, 0)])
    assert n == 0
    n, m, pm = run(oracle, [(100, 100, 0, 0)], [(199.5, 100, 0, 0)])
    assert n == 1
    # the window is centred on vbPrevMatched, not on the keypoint
    n, m, pm = run(oracle, [(100, 100, 0, 0)], [(400, 300, 0, 0)], prev=[(350, 250)])
    assert n == 1 and list(m) == [0]


def test_thresholds_and_previous_position_update(oracle):
    # 50 <= TH_LOW passes, 51 does not (:482); a single candidate leaves bestDist2 at INT_MAX
    assert run(oracle, [(100, 100, 0, 0)], [(110, 100, 0, 50)])[0] == 1
    assert run(oracle, [(100, 100, 0, 0)], [(110, 100, 0, 51)])[0] == 0
    # ratio 0.9: best 36, second 40: 36 < 36.0 is false; best 35 passes (:484)
    assert run(oracle, [(100, 100, 0, 0)], [(110, 100, 0, 36), (120, 100, 0, 40)])[0] == 0
    n, m, pm = run(oracle, [(100, 100, 0, 0), (300, 300, 0, 0)], [(110, 100, 0, 40), (120, 104, 0, 35)])
    assert n == 1 and list(m) == [1, -1]
    # vbPrevMatched of matched keypoints moves to the matched feature, the others stay (:544-546)
    assert pm[0].tolist() == [120.0, 104.0] and pm[1].tolist() == [300.0, 300.0]


def test_a_better_claim_takes_the_feature_over(oracle):
    # keypoint 0 takes feature 0 at distance 20; keypoint 1 is at distance 5 from it: vMatchedDistance = 20 > 5, so
    # it is still a candidate (:465), the match moves over and nmatches stays at 1 (:486-495)
    f2 = [(100, 100, 0, 0)]
    n, m, pm = run(oracle, [(100, 100, 0, 20), (101, 100, 0, 5)], f2)
    assert n == 1 and list(m) == [-1, 0]
    # an equal or larger distance does not see the feature any more (<=, :465)
    n, m, pm = run(oracle, [(100, 100, 0, 20), (101, 100, 0, 20)], f2)
    assert n == 1 and list(m) == [0, -1]
    n, m, pm = run(oracle, [(100, 100, 0, 20), (101, 100, 0, 30)], f2)
    assert n == 1 and list(m) == [0, -1]
    # a held feature drops out of a later keypoint's second best as well.  Keypoint 0 (zeros): distances 2 and 21 ->
    # takes feature 0.  Keypoint 1 (ten ones): distance 8 to feature 0 (held at 2 <= 8: invisible) and 11 to
    # feature 1, now the only candidate -> match; with feature 0 visible 11 < 0.9 * 8 would have failed
    f2 = [(100, 100, 0, 2), (105, 100, 0, 21)]
    n, m, pm = run(oracle, [(100, 100, 0, 0), (101, 100, 0, 10)], f2)
    assert n == 2 and list(m) == [0, 1]


def test_rotation_histogram_counts_replaced_matches(oracle):
    # eleven pairs rotate by 0 degrees, one by 90: 1 < 0.1 * 11 -> only bin 0 stays (:515-540)
    f1 = [(50 + 40 * i, 50, 0, 0, 0.0) for i in range(12)]
    f2 = [(50 + 40 * i, 50, 0, 0, 0.0) for i in range(12)]
    f1[11] = (50 + 40 * 11, 50, 0, 0, 90.0)
    n, m, pm = run(oracle, f1, f2, window=10, check=True)
    assert n == 11 and m[11] == -1 and list(m[:11]) == list(range(11))
    n, m, pm = run(oracle, f1, f2, window=10, check=False)
    assert n == 12
    # ten at 0 degrees and one at 90 whose match is taken over by a twelfth keypoint that rotates by 0: the replaced
    # entry still counts in bin 3 (rotHist keeps it), so bin 3 holds 1 and bin 0 holds 11 -> 1 < 1.1: bin 3 goes,
    # but its only entry is already -1 and nmatches is not decremented twice (:532-536)
    f1 = [(50 + 40 * i, 50, 0, 0, 0.0) for i in range(10)] + [(500, 300, 0, 20, 90.0), (501, 300, 0, 5, 0.0)]
    f2 = [(50 + 40 * i, 50, 0, 0, 0.0) for i in range(10)] + [(500, 300, 0, 0, 0.0)]
    n, m, pm = run(oracle, f1, f2, window=10, check=True)
    assert n == 11 and m[10] == -1 and m[11] == 10
    # nine at 0 degrees plus the replaced one at 90: 1 < 0.1 * 10 is false -> bin 3 stays; nothing changes either way
    n, m, pm = run(oracle, f1[1:], f2[1:], window=10, check=True)
    assert n == 10
    # the replaced entry decides which bins stay: bin 0 holds 10, bins 3, 6 and 9 one live match each, bin 9 one
    # replaced entry on top (2) -> the three maxima are bins 0, 9 and 3 and the match of bin 6 goes; had the replaced
    # entry not counted, bins 0, 3 and 6 would have stayed and the match of bin 9 gone
    f1 = [(50 + 40 * i, 50, 0, 0, 0.0) for i in range(10)]
    f2 = [(50 + 40 * i, 50, 0, 0, 0.0) for i in range(10)]
    f1 += [(100, 300, 0, 0, 90.0), (200, 300, 0, 0, 180.0), (300, 300, 0, 20, 270.0), (301, 300, 0, 5, 270.0)]
    f2 += [(100, 300, 0, 0, 0.0), (200, 300, 0, 0, 0.0), (300, 300, 0, 0, 0.0)]
    n, m, pm = run(oracle, f1, f2, window=10, check=True)
    assert n == 12 and list(m[10:]) == [10, -1, -1, 12]


def test_empty_sides(oracle):
    n, m, pm = run(oracle, [], [(100, 100, 0, 0)])
    assert n == 0
    n, m, pm = run(oracle, [(100, 100, 0, 0)], [])
    assert n == 0 and list(m) == [-1] and pm[0].tolist() == [100.0, 100.0]
