"""Known-answer tests of the oracle's DBoW2 L1Scoring::score (reference Thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-68)."""
import numpy as np


def test_hand_cases(oracle):
    # a vector against itself: every word is common, |0| - 2|vi| summed = -2, score = 1 for an L1-normalised vector
    ids, vals = [3, 7, 20], [0.5, 0.25, 0.25]
    assert oracle.bow_score_l1(ids, vals, ids, vals) == 1.0
    # no common word: 0 (and -0/2 stays a zero)
    assert oracle.bow_score_l1([1, 2], [0.5, 0.5], [3, 4], [0.5, 0.5]) == 0.0
    # one common word: |0.5 - 0.25| - 0.5 - 0.25 = -0.5 -> 0.25
    assert oracle.bow_score_l1([1, 2], [0.5, 0.5], [2, 3], [0.25, 0.75]) == 0.25
    # empty vectors
    assert oracle.bow_score_l1([], [], [1], [1.0]) == 0.0
    assert oracle.bow_score_l1([1], [1.0], [], []) == 0.0


def test_accumulation_order_is_ascending_word_id(oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        ids1 = np.sort(rng.choice(5000, 700, replace=False)).astype(np.uint32)
        ids2 = np.sort(rng.choice(5000, 900, replace=False)).astype(np.uint32)
        v1, v2 = rng.random(700), rng.random(900)
        v1 /= v1.sum()
        v2 /= v2.sum()
        second = dict(zip(ids2.tolist(), v2.tolist()))
        s = 0.0
        for w, vi in zip(ids1.tolist(), v1.tolist()):       # Python floats are doubles: the reference's sum, term by term
            if w in second:
                wi = second[w]
                s += abs(vi - wi) - abs(vi) - abs(wi)
        assert oracle.bow_score_l1(ids1, v1, ids2, v2) == -s / 2.0
