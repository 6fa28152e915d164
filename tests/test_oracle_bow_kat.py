"""Known-answer tests of the oracle's bag-of-words transform (Frame::ComputeBoW = DBoW2's
TemplatedVocabulary::transform, reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1260)
against hand-built trees and an independent numpy restatement, and of the library's vocabulary
loader (text format of loadFromTextFile, :1338-1425).  The ORB-SLAM2 vocabulary file is not part
of the reference checkout, so no result of the real ORBvoc.txt is pinned here."""
import numpy as np
import pytest

import orb_slam2_amd as oa


def popcount(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def descend(voc, d, levelsup):
    """Independent restatement of :1218-1260 on the Vocabulary's arrays."""
    children = {}
    for i in range(1, len(voc.parent)):
        children.setdefault(int(voc.parent[i]), []).append(i)
    words, nw = {}, 0
    for i in range(1, len(voc.parent)):
        if voc.is_leaf[i]:
            words[i] = nw
            nw += 1
    cur, level, nid = 0, 0, 0
    while cur in children:
        ch = children[cur]
        dist = [popcount(d, voc.desc[c]) for c in ch]
        cur = ch[int(np.argmin(dist))]  # argmin returns the first minimum
        level += 1
        if level == voc.L - levelsup:
            nid = cur
    return words.get(cur, 0), nid, float(voc.weight[cur])


def tiny_tree(oracle, scoring=0, weighting=0):
    # root -> 1, 2 ; 1 -> 3, 4 ; 2 -> 5, 6 (k = 2, L = 2)
    parent = [0, 0, 0, 1, 1, 2, 2]
    is_leaf = [0, 0, 0, 1, 1, 1, 1]
    desc = np.zeros((7, 32), np.uint8)
    desc[1, :] = 0x00
    desc[2, :] = 0xFF
    desc[3, 0] = 0x00
    desc[4, 0] = 0x0F
    desc[5, :] = 0xFF
    desc[6, :] = 0xFF
    desc[6, 0] = 0xF0
    weight = [0, 0, 0, 1.5, 2.5, 0.0, 4.0]  # word 2 (node 5) is a stopped word
    return oracle.Vocabulary(2, 2, scoring, weighting, parent, is_leaf, desc, weight)


def test_descent_ties_and_levels(oracle):
    v = tiny_tree(oracle)
    assert v.n_words == 4
    z = np.zeros(32, np.uint8)
    assert v.transform_feature(z, 0) == (0, 3, 1.5)          # nid level 2: the word's own node
    assert v.transform_feature(z, 1) == (0, 1, 1.5)          # one level up
    assert v.transform_feature(z, 2) == (0, 0, 1.5)          # level 0 -> root
    assert v.transform_feature(z, 5) == (0, 0, 1.5)
    f = np.zeros(32, np.uint8)
    f[0] = 0x03                                              # distance 2 to node 3 and to node 4: first child wins
    assert v.transform_feature(f, 0) == (0, 3, 1.5)
    f[0] = 0x07
    assert v.transform_feature(f, 0) == (1, 4, 2.5)
    half = np.zeros(32, np.uint8)
    half[:16] = 0xFF                                         # 128 bits from either child of the root: node 1 wins
    assert v.transform_feature(half, 1)[1] == 1
    ones = np.full(32, 0xFF, np.uint8)
    assert v.transform_feature(ones, 1) == (2, 2, 0.0)       # equal children 5 and 6 differ in byte 0 only


def test_bow_vector_sum_order_and_normalisation(oracle):
    v = tiny_tree(oracle)
    z = np.zeros(32, np.uint8)
    f4 = z.copy()
    f4[0] = 0x0F
    ones = np.full(32, 0xFF, np.uint8)
    f6 = ones.copy()
    f6[0] = 0xF0
    feats = np.stack([f4, z, ones, f6, z, f4, z])            # words 1, 0, 2(stopped), 3, 0, 1, 0
    bid, bval, fnode, fidx = v.transform(feats, 1)
    assert list(bid) == [0, 1, 3]
    raw = np.array([1.5 + 1.5 + 1.5, 2.5 + 2.5, 4.0])
    norm = (raw[0] + raw[1]) + raw[2]
    assert (bval == raw / norm).all()                        # L1, accumulated in word order
    assert list(fnode) == [1, 1, 1, 1, 1, 2]                 # the stopped feature (index 2) is in neither vector
    assert list(fidx) == [0, 1, 4, 5, 6, 3]


@pytest.mark.parametrize("scoring,weighting", [(0, 0), (1, 0), (5, 1), (0, 2), (5, 3), (2, 1)])
def test_weighting_and_scoring_types(oracle, scoring, weighting):
    v = tiny_tree(oracle, scoring, weighting)
    z = np.zeros(32, np.uint8)
    f4 = z.copy()
    f4[0] = 0x0F
    feats = np.stack([z, f4, z])
    bid, bval, _, _ = v.transform(feats, 0)
    assert list(bid) == [0, 1]
    raw = np.array([3.0, 2.5]) if weighting in (0, 1) else np.array([1.5, 2.5])   # addWeight / addIfNotExist
    if scoring == 5:                                                              # DOT_PRODUCT: no norm
        exp = raw / 2.0 if weighting in (0, 1) else raw
    elif scoring == 1:
        exp = raw / np.sqrt(raw[0] * raw[0] + raw[1] * raw[1])
    else:
        exp = raw / (raw[0] + raw[1])
    assert (bval == exp).all()


@pytest.mark.parametrize("k,L,prune,stop,seed", [(10, 3, 0.0, 0.0, 1), (6, 4, 0.2, 0.1, 2), (3, 6, 0.1, 0.05, 3)])
def test_random_trees_against_numpy_restatement(oracle, k, L, prune, stop, seed):
    v = oracle.random_vocabulary(k, L, seed, prune=prune, stop=stop)
    rng = np.random.default_rng(seed + 100)
    feats = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    feats[50:100] = v.desc[rng.integers(1, len(v.desc), 50)]  # exact centres: distance 0 somewhere down the tree
    for levelsup in (0, 2, 4):
        got = [v.transform_feature(f, levelsup) for f in feats]
        exp = [descend(v, f, levelsup) for f in feats]
        assert got == exp
    bid, bval, fnode, fidx = v.transform(feats, 2)
    acc, fv = {}, {}
    for i, f in enumerate(feats):
        w, nid, wt = descend(v, f, 2)
        if wt > 0:
            acc[w] = acc.get(w, 0.0) + wt if w in acc else wt
            fv.setdefault(nid, []).append(i)
    ids = sorted(acc)
    norm = 0.0
    for i in ids:
        norm += abs(acc[i])
    assert list(bid) == ids
    assert (bval == np.array([acc[i] / norm for i in ids])).all()
    assert list(fnode) == [n for n in sorted(fv) for _ in fv[n]]
    assert list(fidx) == [i for n in sorted(fv) for i in fv[n]]


def test_empty_inputs(oracle):
    v = tiny_tree(oracle)
    bid, bval, fnode, fidx = v.transform(np.zeros((0, 32), np.uint8), 4)
    assert len(bid) == 0 and len(fnode) == 0
    root_only = oracle.Vocabulary(10, 6, 0, 0, [0], [0], np.zeros((1, 32), np.uint8), [0.0])  # empty(): nothing produced
    bid, _, fnode, _ = root_only.transform(np.zeros((3, 32), np.uint8), 4)
    assert len(bid) == 0 and len(fnode) == 0


def test_library_loader_round_trip(oracle, tmp_path):
    v = oracle.random_vocabulary(5, 3, 7, prune=0.2, stop=0.1)
    path = tmp_path / "voc.txt"
    v.save_text(path)
    a = oa.ORBVocabulary(5, 3, 0, 0, v.parent, v.is_leaf, v.desc, v.weight)
    b = oa.ORBVocabulary.load_text(path)
    assert a.info() == b.info() == {"k": 5, "L": 3, "n_nodes": len(v.parent), "n_words": v.n_words}
    for i in range(len(v.parent)):
        na, nb = a.node(i), b.node(i)
        assert na[:2] == nb[:2] and na[3:] == nb[3:]
        if i:  # the root has no line in the file: its centre is never used
            assert (na[2] == nb[2]).all() and (na[2] == v.desc[i]).all() and na[3] == v.weight[i]
    with pytest.raises(oa.OrbxError):
        oa.ORBVocabulary.load_text(tmp_path / "missing.txt")
    with pytest.raises(oa.OrbxError):
        oa.ORBVocabulary(5, 3, 0, 0, [0, 2, 0], [0, 1, 1], np.zeros((3, 32), np.uint8), [0, 1, 1])  # parent after child
    a.close()
    b.close()
