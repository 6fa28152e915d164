"""Known-answer tests of the oracle's restatements of ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...)
(reference src/ORBmatcher.cc:159-290) and ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...) (:551-684): every
case is small enough to be worked out by hand from the reference's loops."""
import numpy as np


def desc_with_bits(n_bits, base=None):
    """A 32-byte descriptor that differs from `base` (default: zeros) in exactly n_bits bits."""
    d = np.zeros(32, np.uint8) if base is None else base.copy()
    for b in range(n_bits):
        d[b // 8] ^= np.uint8(1 << (b % 8))
    return d


def run(oracle, kf, f, nn_ratio=0.7, check=False):
    """kf / f: lists of (node, descriptor, angle[, valid]) per feature, feature index = list position."""
    def side(feats):
        node = np.array([x[0] for x in feats], np.uint32)
        order = np.lexsort((np.arange(len(feats)), node))     # by node, then feature index
        desc = np.stack([x[1] for x in feats]) if feats else np.zeros((0, 32), np.uint8)
        ang = np.array([x[2] for x in feats], np.float32)
        valid = np.array([x[3] if len(x) > 3 else 1 for x in feats], np.uint8)
        return node[order], order.astype(np.uint32), desc, ang, valid
    nk, ik, dk, ak, vk = side(kf)
    nf, if_, df, af, _ = side(f)
    return oracle.search_by_bow(nk, ik, dk, ak, vk, nf, if_, df, af, nn_ratio, check)


def test_same_node_only_and_thresholds(oracle):
    z = np.zeros(32, np.uint8)
    # keyframe: feature 0 in node 5, feature 1 in node 9 (no such node in the frame), feature 2 in node 7
    kf = [(5, z, 0.0), (9, z, 0.0), (7, z, 0.0)]
    # frame: node 5 holds features 0 (distance 50) and 1 (distance 200); node 7 holds feature 2 (distance 51);
    # node 6 holds feature 3 (distance 0, but no keyframe feature looks into node 6)
    f = [(5, desc_with_bits(50), 0.0), (5, desc_with_bits(200), 0.0), (7, desc_with_bits(51), 0.0), (6, z, 0.0)]
    n, m = run(oracle, kf, f)
    assert n == 1 and list(m) == [0, -1, -1, -1]        # 50 <= TH_LOW passes, 51 does not (:228)


def test_ratio_test_and_single_candidate(oracle):
    z = np.zeros(32, np.uint8)
    kf = [(1, z, 0.0)]
    # best 30, second 40: 30 < 0.7 * 40 = 28 is false -> no match (:230)
    n, m = run(oracle, kf, [(1, desc_with_bits(30), 0.0), (1, desc_with_bits(40), 0.0)])
    assert n == 0 and list(m) == [-1, -1]
    # best 27, second 40: 27 < 28 -> match
    n, m = run(oracle, kf, [(1, desc_with_bits(40), 0.0), (1, desc_with_bits(27), 0.0)])
    assert n == 1 and list(m) == [-1, 0]
    # a single candidate: the second distance stays 256, 50 < 179.2 -> match
    n, m = run(oracle, kf, [(1, desc_with_bits(50), 0.0)])
    assert n == 1 and list(m) == [0]
    # equal distances: the first of them is the best, the other one the second best (:216-225), 10 < 7 fails
    n, m = run(oracle, kf, [(1, desc_with_bits(10), 0.0), (1, desc_with_bits(10), 0.0)])
    assert n == 0


def test_taken_features_are_skipped_by_later_keyframe_features(oracle):
    z = np.zeros(32, np.uint8)
    a = desc_with_bits(5)
    # both keyframe features are closest to frame feature 0; the first takes it (0 < 0.7 * 40), the second
    # then sees only feature 1 (distance 45 from it, second best 256) and takes that one (:209-210)
    kf = [(3, z, 0.0), (3, a, 0.0)]
    f = [(3, z, 0.0), (3, desc_with_bits(45, a), 0.0)]
    n, m = run(oracle, kf, f)
    assert n == 2 and list(m) == [0, 1]
    # with the order of the keyframe features swapped the other one gets feature 0
    n, m = run(oracle, [(3, a, 0.0), (3, z, 0.0)], [(3, z, 0.0), (3, desc_with_bits(45, a), 0.0)])
    # feature a: distances 5 (to z) and 45 -> 5 < 31.5 -> takes frame feature 0; feature z then sees only
    # frame feature 1 at distance 40 -> match as well
    assert n == 2 and list(m) == [0, 1]


def test_invalid_keyframe_features_and_empty_sides(oracle):
    z = np.zeros(32, np.uint8)
    n, m = run(oracle, [(1, z, 0.0, 0)], [(1, z, 0.0)])       # no map point / bad map point (:193-197)
    assert n == 0 and list(m) == [-1]
    n, m = run(oracle, [], [(1, z, 0.0)])
    assert n == 0 and list(m) == [-1]
    n, m = run(oracle, [(1, z, 0.0)], [])
    assert n == 0 and len(m) == 0


def test_distance_256_is_never_a_candidate(oracle):
    z = np.zeros(32, np.uint8)
    ones = np.full(32, 255, np.uint8)
    # the only candidate is at distance 256: neither dist < bestDist1 nor dist < bestDist2 holds
    n, m = run(oracle, [(1, z, 0.0)], [(1, ones, 0.0)])
    assert n == 0
    # with a second candidate at distance 40 the best is 40 and the second best stays 256
    n, m = run(oracle, [(1, z, 0.0)], [(1, ones, 0.0), (1, desc_with_bits(40), 0.0)])
    assert n == 1 and list(m) == [-1, 0]


def test_rotation_histogram_keeps_the_main_bins(oracle):
    z = np.zeros(32, np.uint8)
    # twelve keyframe features, each alone in its node with an exact twin in the frame; ten rotate by 0
    # degrees (bin 0), one by 90 (bin 3), one by 180 (bin 6): the second and third bins hold 1 < 0.1 * 10?
    # no: 1 < 1.0 is false, so all three bins stay
    kf, f = [], []
    for i in range(12):
        rot = 0.0 if i < 10 else (90.0 if i == 10 else 180.0)
        kf.append((10 + i, z, rot))
        f.append((10 + i, z, 0.0))
    n, m = run(oracle, kf, f, check=True)
    assert n == 12
    # eleven at 0 degrees: 1 < 0.1 * 11 = 1.1 -> only the first bin stays (:1638-1679)
    kf.append((30, z, 0.0))
    f.append((30, z, 0.0))
    n, m = run(oracle, kf, f, check=True)
    assert n == 11 and m[10] == -1 and m[11] == -1 and m[12] == 12
    # a negative difference wraps: 10 - 350 = -340 -> 20 degrees -> bin round(20 / 30) = 1
    n, m = run(oracle, [(1, z, 10.0), (2, z, 0.0), (3, z, 0.0)], [(1, z, 350.0), (2, z, 0.0), (3, z, 0.0)], check=True)
    assert n == 3                                          # bins 0 (two) and 1 (one): 1 < 0.2 is false
    # without the check nothing is removed
    n, m = run(oracle, kf, f, check=False)
    assert n == 13


def run_kf(oracle, kf1, kf2, nn_ratio=0.7, check=False):
    """kf1 / kf2: lists of (node, descriptor, angle[, valid]) per feature."""
    def side(feats):
        node = np.array([x[0] for x in feats], np.uint32)
        order = np.lexsort((np.arange(len(feats)), node))
        desc = np.stack([x[1] for x in feats]) if feats else np.zeros((0, 32), np.uint8)
        ang = np.array([x[2] for x in feats], np.float32)
        valid = np.array([x[3] if len(x) > 3 else 1 for x in feats], np.uint8)
        return node[order], order.astype(np.uint32), desc, ang, valid
    a, b = side(kf1), side(kf2)
    return oracle.search_by_bow_keyframes(a[0], a[1], a[2], a[3], a[4], b[0], b[1], b[2], b[3], b[4], nn_ratio, check)


def test_keyframe_pair_differences_from_the_frame_version(oracle):
    z = np.zeros(32, np.uint8)
    # the threshold is strict here: 50 is rejected, 49 accepted (:627)
    n, m = run_kf(oracle, [(1, z, 0.0)], [(1, desc_with_bits(50), 0.0)])
    assert n == 0 and list(m) == [-1]
    n, m = run_kf(oracle, [(1, z, 0.0)], [(1, desc_with_bits(49), 0.0)])
    assert n == 1 and list(m) == [0]
    # the result is indexed by the FIRST keyframe's features and holds the second keyframe's feature (:631)
    n, m = run_kf(oracle, [(2, z, 0.0), (1, z, 0.0)], [(1, desc_with_bits(3), 0.0), (7, z, 0.0), (2, desc_with_bits(1), 0.0)])
    assert n == 2 and list(m) == [2, 0]
    # features of the second keyframe without a good map point are not candidates (:603-609): the twin is
    # invisible, the other candidate at distance 40 is taken with the second best at 256
    n, m = run_kf(oracle, [(1, z, 0.0)], [(1, z, 0.0, 0), (1, desc_with_bits(40), 0.0)])
    assert n == 1 and list(m) == [1]
    # a feature of the second keyframe serves one feature of the first only (vbMatched2, :605, :632)
    n, m = run_kf(oracle, [(1, z, 0.0), (1, z, 0.0)], [(1, z, 0.0)])
    assert n == 1 and list(m) == [0, -1]
    # the rotation check clears entries of the first keyframe (:663-682): eleven at 0 degrees, one at 90
    kf1 = [(10 + i, z, 0.0 if i < 11 else 90.0) for i in range(12)]
    kf2 = [(10 + i, z, 0.0) for i in range(12)]
    n, m = run_kf(oracle, kf1, kf2, check=True)
    assert n == 11 and m[11] == -1 and list(m[:11]) == list(range(11))
