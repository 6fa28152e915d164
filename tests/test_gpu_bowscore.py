"""GPU parity of ORBVocabulary::score for the L1 norm (DBoW2 L1Scoring::score, reference
Thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-68) against the CPU oracle: the doubles agree bit for bit, on the BowVectors
the bag-of-words transform leaves for real extractions and on made-up vectors of every length."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth
from hip_rt import DeviceArray

pytestmark = pytest.mark.gpu


def score_on_device(ex, id1, v1, n1, id2, v2, n2, pairs):
    d = [DeviceArray.from_host(np.ascontiguousarray(a)) for a in (id1, v1, n1, id2, v2, n2, pairs)]
    d_s = DeviceArray((len(pairs),), np.float64, fill_byte=0x55)
    ex.bow_score_l1_batch_device(len(pairs), d[0].ptr, d[1].ptr, d[2].ptr, id1.shape[1], d[3].ptr, d[4].ptr, d[5].ptr, id2.shape[1],
                                 d[6].ptr, d_s.ptr, sync=True)
    out = d_s.to_host()
    for a in d + [d_s]:
        a.free()
    return out


def test_scores_of_extracted_frames(oracle):
    from test_gpu_bow import run_bow
    w, h, nf, n = 640, 480, 1000, 6
    base = synth.make_frame(w + 64, h + 16, 2, 9)
    other = synth.make_frame(w, h, 2, 10)
    frames = np.stack([np.ascontiguousarray(base[2 * i:2 * i + h, 8 * i:8 * i + w]) for i in range(n - 1)] + [other])
    ex = oa.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    d_img = DeviceArray.from_host(frames)
    d_kps, d_desc = DeviceArray((n, cap, 28), np.uint8, fill_byte=0), DeviceArray((n, cap, 32), np.uint8, fill_byte=0)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    ex.extract_batch_device(d_img.ptr, n, w, h, w, h * w, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=True)
    ovoc = oracle.random_vocabulary(10, 4, 21)
    voc = oa.ORBVocabulary(10, 4, 0, 0, ovoc.parent, ovoc.is_leaf, ovoc.desc, ovoc.weight)
    out = run_bow(ex, voc, d_desc, cap, d_n, n, 2)
    bid, bval, bn = out[3], out[4], out[5]
    pairs = np.array([(i, j) for i in range(n) for j in range(n)], np.int32)
    got = score_on_device(ex, bid, bval, bn, bid, bval, bn, pairs)
    exp = np.array([oracle.bow_score_l1(bid[i, :bn[i]], bval[i, :bn[i]], bid[j, :bn[j]], bval[j, :bn[j]]) for i, j in pairs])
    assert (got.view(np.uint64) == exp.view(np.uint64)).all()
    s = got.reshape(n, n)
    assert np.abs(np.diag(s) - 1.0).max() < 1e-12          # a vector against itself
    assert s[0, 1] > s[0, n - 1]                             # the next view of the scene scores above another scene
    voc.close()
    ex.close()
    for a in (d_img, d_kps, d_desc, d_n):
        a.free()


def test_scores_of_made_up_vectors(oracle):
    rng = np.random.default_rng(8)
    ex = oa.ORBextractor(500, 1.2, 4, 20, 7, max_width=320, max_height=240, max_batch=1, keep_pyramid=False)
    m, cap_1, cap_2 = 24, 2048, 700
    n1 = np.array([0, 1, 63, 64, 65, 127, 128, 129, 500, 1000, 2047, 2048] * 2, np.int32)
    n2 = rng.integers(0, cap_2 + 1, m).astype(np.int32)
    n2[:3] = [0, 1, cap_2]
    id1, id2 = np.zeros((m, cap_1), np.uint32), np.zeros((m, cap_2), np.uint32)
    v1, v2 = np.zeros((m, cap_1)), np.zeros((m, cap_2))
    for i in range(m):
        id1[i, :n1[i]] = np.sort(rng.choice(3000, n1[i], replace=False))
        id2[i, :n2[i]] = np.sort(rng.choice(3000, n2[i], replace=False))
        v1[i, :n1[i]] = rng.random(n1[i]) / max(1, n1[i])
        v2[i, :n2[i]] = rng.random(n2[i]) / max(1, n2[i])
    pairs = np.array([(i, j) for i in range(m) for j in range(0, m, 3)], np.int32)
    got = score_on_device(ex, id1, v1, n1, id2, v2, n2, pairs)
    exp = np.array([oracle.bow_score_l1(id1[i, :n1[i]], v1[i, :n1[i]], id2[j, :n2[j]], v2[j, :n2[j]]) for i, j in pairs])
    assert (got.view(np.uint64) == exp.view(np.uint64)).all()
    with pytest.raises(oa.OrbxError):
        ex.bow_score_l1_batch_device(1, 16, 16, 16, 100, 16, 16, 16, 100, 0, 16)
    ex.close()
