"""Known-answer tests of the oracle's ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:686-852, with
CheckDistEpipolarLine, :140-157): cases small enough to be worked out by hand from the reference's loops."""
import numpy as np

import oracle_lib

SCALE = np.array([1.2 ** i for i in range(8)], np.float32)
SIGMA2 = (SCALE * SCALE).astype(np.float32)
# x1' F12 = [0, 1, -y1]: the epipolar line of (x1, y1) is the row y = y1 (a camera moved along x)
F_ROWS = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
FAR = (-1e4, -1e4)         # an epipole far outside the image


def ones(n):
    return np.packbits(np.arange(256) < n)


def run(kf1, kf2, f12=F_ROWS, epipole=FAR, only_stereo=False, check=False):
    """kf: list of dicts per feature, feature index = list position: node, bits (or desc), x, y, and optionally octave,
    angle, mp (holds a map point), ur (mvuRight, default -1)."""
    def side(feats):
        node = np.array([p["node"] for p in feats], np.uint32)
        order = np.lexsort((np.arange(len(feats)), node))
        desc = np.stack([p["desc"] if "desc" in p else ones(p["bits"]) for p in feats]) if feats else np.zeros((0, 32), np.uint8)
        xy = np.array([(p["x"], p["y"]) for p in feats], np.float32).reshape(-1, 2)
        angle = np.array([p.get("angle", 0.0) for p in feats], np.float32)
        octave = np.array([p.get("octave", 0) for p in feats], np.int32)
        mp = np.array([p.get("mp", 0) for p in feats], np.uint8)
        ur = np.array([p.get("ur", -1.0) for p in feats], np.float32)
        return node[order], order.astype(np.uint32), desc, xy, angle, octave, mp, ur
    return oracle_lib.search_for_triangulation(side(kf1), side(kf2), f12, epipole, SCALE, SIGMA2, only_stereo, check)


def P(node, bits, x, y, **kw):
    return dict(node=node, bits=bits, x=x, y=y, **kw)


def test_threshold_ties_and_shared_features(oracle):
    # 50 <= TH_LOW passes, 51 does not (:767)
    assert run([P(1, 0, 100, 100)], [P(1, 50, 300, 100)])[0] == 1
    assert run([P(1, 0, 100, 100)], [P(1, 51, 300, 100)])[0] == 0
    # the smallest distance wins; of equal distances the LAST in the node's list (dist > bestDist skips, equal does not)
    n, m = run([P(1, 0, 100, 100)], [P(1, 30, 300, 100), P(1, 10, 310, 100), P(1, 10, 320, 100), P(1, 20, 330, 100)])
    assert n == 1 and list(m) == [2]
    # vbMatched2 is never set: one feature of the second keyframe serves both features of the first
    n, m = run([P(1, 0, 100, 100), P(1, 3, 120, 100)], [P(1, 1, 300, 100)])
    assert n == 2 and list(m) == [0, 0]
    # other nodes are not looked at, nodes on one side only are skipped
    n, m = run([P(1, 0, 100, 100), P(5, 0, 100, 100)], [P(2, 0, 300, 100), P(5, 7, 300, 100), P(9, 0, 300, 100)])
    assert n == 1 and list(m) == [-1, 1]


def test_map_points_and_stereo_only(oracle):
    # features that hold a map point are skipped on both sides (:728-732, :751-755)
    assert run([P(1, 0, 100, 100, mp=1)], [P(1, 0, 300, 100)])[0] == 0
    n, m = run([P(1, 0, 100, 100)], [P(1, 0, 300, 100, mp=1), P(1, 40, 310, 100)])
    assert n == 1 and list(m) == [1]
    # bOnlyStereo: both features need mvuRight >= 0 (:736-738, :759-761); 0 counts as stereo
    assert run([P(1, 0, 100, 100)], [P(1, 0, 300, 100, ur=250.0)], only_stereo=True)[0] == 0
    assert run([P(1, 0, 100, 100, ur=0.0)], [P(1, 0, 300, 100)], only_stereo=True)[0] == 0
    n, m = run([P(1, 0, 100, 100, ur=0.0)], [P(1, 0, 300, 100), P(1, 9, 310, 100, ur=280.0)], only_stereo=True)
    assert n == 1 and list(m) == [1]


def test_epipole_distance_only_for_monocular_pairs(oracle):
    # distex^2 + distey^2 < 100 * mvScaleFactors[octave] rejects (:772-778): 9.9 px from the epipole at level 0
    ep = (300.0, 100.0)
    assert run([P(1, 0, 100, 100)], [P(1, 0, 309.9, 100)], epipole=ep)[0] == 0
    assert run([P(1, 0, 100, 100)], [P(1, 0, 310.0, 100)], epipole=ep)[0] == 1          # 100 < 100 is false
    # at level 2 the limit is 100 * 1.44: 11.9 px is too close, 12.1 px is not
    assert run([P(1, 0, 100, 100)], [P(1, 0, 311.9, 100, octave=2)], epipole=ep)[0] == 0
    assert run([P(1, 0, 100, 100)], [P(1, 0, 312.1, 100, octave=2)], epipole=ep)[0] == 1
    # with a stereo measurement on either side the epipole does not matter
    assert run([P(1, 0, 100, 100, ur=90.0)], [P(1, 0, 301, 100)], epipole=ep)[0] == 1
    assert run([P(1, 0, 100, 100)], [P(1, 0, 301, 100, ur=290.0)], epipole=ep)[0] == 1


def test_epipolar_line_distance(oracle):
    # dsqr = (y2 - y1)^2 against 3.84 * mvLevelSigma2[octave of kp2] (:154-156): 1.95 px passes at level 0, 1.97 px not
    assert run([P(1, 0, 100, 100)], [P(1, 0, 300, 101.95)])[0] == 1
    assert run([P(1, 0, 100, 100)], [P(1, 0, 300, 101.97)])[0] == 0
    # level 1: 3.84 * 1.44 = 5.5296 -> 2.35 px passes, 2.36 px not
    assert run([P(1, 0, 100, 100)], [P(1, 0, 300, 102.35, octave=1)])[0] == 1
    assert run([P(1, 0, 100, 100)], [P(1, 0, 300, 102.36, octave=1)])[0] == 0
    # a zero line (den == 0) matches nothing (:151-152)
    assert run([P(1, 0, 100, 100)], [P(1, 0, 300, 100)], f12=np.zeros((3, 3), np.float32))[0] == 0
    # a closer descriptor off the line does not lower bestDist: the farther one on the line is taken (:780-784)
    n, m = run([P(1, 0, 100, 100)], [P(1, 5, 300, 150), P(1, 30, 310, 100)])
    assert n == 1 and list(m) == [1]
    # a general matrix: l = x1' F12 with F12 = [[0, 0, 0.01], [0, 0, -1], [-1, 1, 0]]: for (100, 100): a = -1, b = 1,
    # c = 1 - 100 = -99 -> the line -x + y - 99 = 0; (200, 299) lies on it, (200, 302) is 3/sqrt(2) = 2.12 px away
    F = np.array([[0, 0, 0.01], [0, 0, -1], [-1, 1, 0]], np.float32)
    assert run([P(1, 0, 100, 100)], [P(1, 0, 200, 299)], f12=F)[0] == 1
    assert run([P(1, 0, 100, 100)], [P(1, 0, 200, 302)], f12=F)[0] == 0
    assert run([P(1, 0, 100, 100)], [P(1, 0, 200, 301.7)], f12=F)[0] == 1     # 2.7 / sqrt(2) = 1.91 px


def test_rotation_histogram(oracle):
    # eleven pairs rotate by 0 degrees, one by 90: 1 < 0.1 * 11 -> only bin 0 stays; removal is unconditional (:828-837)
    kf1 = [P(10 + i, 0, 100, 100 + i, angle=0.0 if i < 11 else 90.0) for i in range(12)]
    kf2 = [P(10 + i, 0, 300, 100 + i) for i in range(12)]
    n, m = run(kf1, kf2, check=True)
    assert n == 11 and m[11] == -1 and list(m[:11]) == list(range(11))
    assert run(kf1, kf2, check=False)[0] == 12


def test_empty_sides(oracle):
    n, m = run([], [P(1, 0, 300, 100)])
    assert n == 0
    n, m = run([P(1, 0, 100, 100)], [])
    assert n == 0 and list(m) == [-1]
