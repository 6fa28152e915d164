"""GPU parity of the bag-of-words stage (Frame::ComputeBoW, reference src/Frame.cc:411-418 =
DBoW2's transform, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1260) against the CPU oracle:
word ids, node ids, BowVector values (float64, bit-exact: both sides add and divide in the
reference's order) and the FeatureVector, on seeded random vocabulary trees (the ORB-SLAM2
vocabulary file is not part of the reference checkout: parity unpinned against real ORBvoc.txt)."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth

pytestmark = pytest.mark.gpu


def run_bow(ex, voc, d_desc, cap, d_n, n, levelsup):
    from hip_rt import DeviceArray
    d_word = DeviceArray((n, cap), np.uint32, fill_byte=0x55)
    d_node = DeviceArray((n, cap), np.uint32, fill_byte=0x55)
    d_w = DeviceArray((n, cap), np.float64, fill_byte=0x55)
    d_bid = DeviceArray((n, cap), np.uint32, fill_byte=0x55)
    d_bval = DeviceArray((n, cap), np.float64, fill_byte=0x55)
    d_bn = DeviceArray((n,), np.int32, fill_byte=0x55)
    d_fnode = DeviceArray((n, cap), np.uint32, fill_byte=0x55)
    d_fidx = DeviceArray((n, cap), np.uint32, fill_byte=0x55)
    d_fn = DeviceArray((n,), np.int32, fill_byte=0x55)
    ex.bow_transform_batch_device(voc, n, d_desc.ptr, cap, d_n.ptr, levelsup, d_word.ptr, d_node.ptr, d_w.ptr, d_bid.ptr,
                                  d_bval.ptr, d_bn.ptr, d_fnode.ptr, d_fidx.ptr, d_fn.ptr, sync=True)
    arrays = (d_word, d_node, d_w, d_bid, d_bval, d_bn, d_fnode, d_fidx, d_fn)
    out = [a.to_host() for a in arrays]
    for a in arrays:
        a.free()
    return out


def check(oracle, ovoc, descs, counts, out, levelsup):
    word, node, w, bid, bval, bn, fnode, fidx, fn = out
    for f, cnt in enumerate(counts):
        exp = [ovoc.transform_feature(d, levelsup) for d in descs[f, :cnt]]
        assert list(word[f, :cnt]) == [e[0] for e in exp], "words of frame %d" % f
        assert list(node[f, :cnt]) == [e[1] for e in exp], "nodes of frame %d" % f
        assert (w[f, :cnt].view(np.uint64) == np.array([e[2] for e in exp]).view(np.uint64)).all()
        obid, obval, ofnode, ofidx = ovoc.transform(descs[f, :cnt], levelsup)
        assert bn[f] == len(obid) and fn[f] == len(ofnode)
        assert (bid[f, :bn[f]] == obid).all()
        assert (bval[f, :bn[f]].view(np.uint64) == obval.view(np.uint64)).all(), "BowVector values of frame %d" % f
        assert (fnode[f, :fn[f]] == ofnode).all() and (fidx[f, :fn[f]] == ofidx).all()


@pytest.mark.parametrize("k,L,prune,stop,scoring,weighting,levelsup,seed", [
    (10, 3, 0.0, 0.0, 0, 0, 1, 11),    # complete tree, ORB-SLAM2's L1_NORM + TF_IDF
    (10, 4, 0.15, 0.1, 0, 0, 2, 12),   # inner nodes that end early, stopped words
    (7, 5, 0.1, 0.05, 1, 0, 4, 13),    # L2 norm; levelsup near L
    (4, 6, 0.1, 0.0, 5, 1, 9, 14),     # DOT_PRODUCT + TF: values divided by the word count; levelsup > L
    (12, 2, 0.0, 0.2, 0, 3, 0, 15),    # BINARY: first weight kept; more children than lanes of a group share
    (20, 2, 0.0, 0.0, 3, 2, 1, 16),    # k = 20 (the loader's limit), IDF
])
def test_bow_after_extraction(oracle, k, L, prune, stop, scoring, weighting, levelsup, seed):
    from hip_rt import DeviceArray
    w, h, nf, n = 640, 480, 1000, 3
    frames = np.stack([synth.make_frame(w, h, 2, seed + i) for i in range(n)])
    frames[2] = 90                                     # a flat frame: no keypoints, empty vectors
    ex = oa.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    d_frames = DeviceArray.from_host(frames)
    d_kps = DeviceArray((n, cap, 28), np.uint8)
    d_desc = DeviceArray((n, cap, 32), np.uint8)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    ex.extract_batch_device(d_frames.ptr, n, w, h, w, w * h, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=True)
    counts, descs = d_n.to_host(), d_desc.to_host()
    assert counts[0] > 500 and counts[2] == 0
    ovoc = oracle.random_vocabulary(k, L, seed, scoring=scoring, weighting=weighting, prune=prune, stop=stop)
    voc = oa.ORBVocabulary(k, L, scoring, weighting, ovoc.parent, ovoc.is_leaf, ovoc.desc, ovoc.weight)
    out = run_bow(ex, voc, d_desc, cap, d_n, n, levelsup)
    check(oracle, ovoc, descs, counts, out, levelsup)
    voc.close()
    ex.close()
    for a in (d_frames, d_kps, d_desc, d_n):
        a.free()


def test_bow_repeated_words_and_capacity(oracle):
    """Many features per word (a small vocabulary), descriptors that equal node centres, a second call on
    the same vocabulary, and the capacity / argument errors."""
    from hip_rt import DeviceArray
    rng = np.random.default_rng(5)
    ovoc = oracle.random_vocabulary(3, 2, 21, stop=0.2)          # 9 words for 1500 features
    voc = oa.ORBVocabulary(3, 2, 0, 0, ovoc.parent, ovoc.is_leaf, ovoc.desc, ovoc.weight)
    n, cap = 2, 1500
    descs = rng.integers(0, 256, (n, cap, 32), dtype=np.uint8)
    descs[0, :400] = ovoc.desc[rng.integers(1, len(ovoc.desc), 400)]
    counts = np.array([1500, 777], np.int32)
    ex = oa.ORBextractor(500, 1.2, 4, 20, 7, max_width=320, max_height=240, max_batch=2, keep_pyramid=False)
    d_desc, d_n = DeviceArray.from_host(descs), DeviceArray.from_host(counts)
    for levelsup in (1, 0):
        out = run_bow(ex, voc, d_desc, cap, d_n, n, levelsup)
        check(oracle, ovoc, descs, counts, out, levelsup)
    with pytest.raises(oa.OrbxError):
        ex.bow_transform_batch_device(voc, n, d_desc.ptr, 20000, d_n.ptr, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1)  # kp_cap > 16384
    # a capacity above 8192: the two key arrays no longer fit the LDS side by side
    big = 9000
    descs2 = rng.integers(0, 256, (1, big, 32), dtype=np.uint8)
    counts2 = np.array([8800], np.int32)
    d_desc2, d_n2 = DeviceArray.from_host(descs2), DeviceArray.from_host(counts2)
    out = run_bow(ex, voc, d_desc2, big, d_n2, 1, 1)
    check(oracle, ovoc, descs2, counts2, out, 1)
    d_desc2.free()
    d_n2.free()
    voc.close()
    ex.close()
    d_desc.free()
    d_n.free()
