"""GPU parity of ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (reference src/ORBmatcher.cc:159-290) against the
CPU oracle's sequential loop: the keyframe feature every frame feature ends with and nmatches, on the chain
extract -> bag of words -> match for frames that show the same scene, and on made-up FeatureVectors with
crowded nodes (hundreds to thousands of candidates per keyframe feature, long chains of taken features)."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth

pytestmark = pytest.mark.gpu


def run_device(ex, kf, f, nn_ratio, check):
    """kf = (fv_node, fv_idx, fv_n, desc, kps, valid) and f = (fv_node, fv_idx, fv_n, desc, kps) as host arrays with a
    leading pair axis; returns (match [n][cap_f], nmatches [n])."""
    from hip_rt import DeviceArray
    n, cap_kf, cap_f = kf[0].shape[0], kf[0].shape[1], f[0].shape[1]
    d = [DeviceArray.from_host(np.ascontiguousarray(a)) for a in kf + f]
    d_match = DeviceArray((n, cap_f), np.int32, fill_byte=0x55)
    d_nm = DeviceArray((n,), np.int32, fill_byte=0x55)
    ex.search_by_bow_batch_device(n, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr, d[5].ptr, cap_kf, d[6].ptr, d[7].ptr,
                                  d[8].ptr, d[9].ptr, d[10].ptr, cap_f, nn_ratio, check, d_match.ptr, d_nm.ptr, sync=True)
    out = d_match.to_host(), d_nm.to_host()
    for a in d + [d_match, d_nm]:
        a.free()
    return out


def run_device_keyframes(ex, k1, k2, nn_ratio, check):
    """k1, k2 = (fv_node, fv_idx, fv_n, desc, kps, valid) with a leading pair axis; returns (match12 [n][cap_1], nmatches)."""
    from hip_rt import DeviceArray
    n, cap_1, cap_2 = k1[0].shape[0], k1[0].shape[1], k2[0].shape[1]
    d = [DeviceArray.from_host(np.ascontiguousarray(a)) for a in k1 + k2]
    d_match = DeviceArray((n, cap_1), np.int32, fill_byte=0x55)
    d_nm = DeviceArray((n,), np.int32, fill_byte=0x55)
    ex.search_by_bow_keyframes_batch_device(n, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr, d[5].ptr, cap_1, d[6].ptr,
                                            d[7].ptr, d[8].ptr, d[9].ptr, d[10].ptr, d[11].ptr, cap_2, nn_ratio, check,
                                            d_match.ptr, d_nm.ptr, sync=True)
    out = d_match.to_host(), d_nm.to_host()
    for a in d + [d_match, d_nm]:
        a.free()
    return out


def check_keyframes_against_oracle(oracle, k1, k2, nn_ratio, check, match, nm):
    total = 0
    for p in range(k1[0].shape[0]):
        n1, n2 = int(k1[2][p]), int(k2[2][p])
        exp_n, exp_m = oracle.search_by_bow_keyframes(k1[0][p, :n1], k1[1][p, :n1], k1[3][p], k1[4][p]["angle"], k1[5][p],
                                                      k2[0][p, :n2], k2[1][p, :n2], k2[3][p], k2[4][p]["angle"], k2[5][p],
                                                      nn_ratio, check)
        assert nm[p] == exp_n, "nmatches of pair %d" % p
        assert (match[p] == exp_m).all(), "matches of pair %d" % p
        total += exp_n
    return total


def check_against_oracle(oracle, kf, f, counts_f, nn_ratio, check, match, nm):
    total = 0
    for p in range(kf[0].shape[0]):
        nk, nf = int(kf[2][p]), int(f[2][p])
        exp_n, exp_m = oracle.search_by_bow(kf[0][p, :nk], kf[1][p, :nk], kf[3][p], kf[4][p]["angle"], kf[5][p],
                                            f[0][p, :nf], f[1][p, :nf], f[3][p], f[4][p]["angle"], nn_ratio, check)
        assert nm[p] == exp_n, "nmatches of pair %d" % p
        assert (match[p] == exp_m).all(), "matches of pair %d" % p
        assert (match[p, counts_f[p]:] == -1).all()
        total += exp_n
    return total


@pytest.mark.parametrize("k,L,levelsup,check,seed", [(10, 3, 1, True, 3), (10, 4, 2, False, 4), (6, 2, 5, True, 5)])
def test_bow_match_after_extraction(oracle, k, L, levelsup, check, seed):
    """Four views of one scene (shifted by a few pixels); pair p matches view p (the keyframe) with view p + 1."""
    from hip_rt import DeviceArray
    from test_gpu_bow import run_bow
    w, h, nf, n = 640, 480, 1000, 4
    base = synth.make_frame(w + 16, h + 16, 2, seed)
    frames = np.stack([np.ascontiguousarray(base[2 * i:2 * i + h, 3 * i:3 * i + w]) for i in range(n)])
    ex = oa.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    d_frames = DeviceArray.from_host(frames)
    d_kps = DeviceArray((n, cap), oa.KP_DTYPE)
    d_desc = DeviceArray((n, cap, 32), np.uint8)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    ex.extract_batch_device(d_frames.ptr, n, w, h, w, w * h, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=True)
    counts, descs, kps = d_n.to_host(), d_desc.to_host(), d_kps.to_host()
    ovoc = oracle.random_vocabulary(k, L, seed)
    voc = oa.ORBVocabulary(k, L, 0, 0, ovoc.parent, ovoc.is_leaf, ovoc.desc, ovoc.weight)
    out = run_bow(ex, voc, d_desc, cap, d_n, n, levelsup)
    fnode, fidx, fn = out[6], out[7], out[8]
    rng = np.random.default_rng(seed)
    valid = (rng.random((n, cap)) < 0.85).astype(np.uint8)
    nxt = np.roll(np.arange(n), -1)
    kf = (fnode, fidx, fn, descs, kps, valid)
    f = (fnode[nxt], fidx[nxt], fn[nxt], descs[nxt], kps[nxt])
    match, nm = run_device(ex, kf, f, 0.7, check)
    total = check_against_oracle(oracle, kf, f, counts[nxt], 0.7, check, match, nm)
    assert total > 100          # the views do share features
    # the same views as two keyframes (the loop closer's overload)
    valid2 = (rng.random((n, cap)) < 0.85).astype(np.uint8)
    k2 = f + (valid2[nxt],)
    match12, nm12 = run_device_keyframes(ex, kf, k2, 0.75, check)
    assert check_keyframes_against_oracle(oracle, kf, k2, 0.75, check, match12, nm12) > 80
    voc.close()
    ex.close()
    for a in (d_frames, d_kps, d_desc, d_n):
        a.free()


def made_up_side(rng, n_feat, cap, n_nodes, desc):
    node_of = rng.integers(1, n_nodes + 1, n_feat).astype(np.uint32)
    keep = rng.random(n_feat) < 0.95                  # stopped words leave features out of the FeatureVector
    idx = np.nonzero(keep)[0]
    order = np.lexsort((idx, node_of[idx]))
    fv_node = np.zeros(cap, np.uint32)
    fv_idx = np.zeros(cap, np.uint32)
    fv_node[:len(idx)] = node_of[idx][order]
    fv_idx[:len(idx)] = idx[order]
    kps = np.zeros(cap, oa.KP_DTYPE)
    kps["angle"][:n_feat] = rng.choice(np.array([0.0, 1.5, 29.9, 45.0, 90.0, 200.0, 359.5], np.float32), n_feat)
    d = np.zeros((cap, 32), np.uint8)
    d[:n_feat] = desc
    return fv_node, fv_idx, np.int32(len(idx)), d, kps


@pytest.mark.parametrize("n_kf,n_f,cap,n_nodes,flip,seed", [
    (1800, 2000, 2048, 40, 12, 1),     # about 50 candidates per node
    (3000, 3500, 4096, 3, 20, 2),      # about 1200 per node: many chunks of 64, long chains
    (500, 9000, 16384, 1, 25, 3),      # one node, more than 4096 candidates (several words of taken flags)
    (64, 64, 64, 64, 0, 4),            # capacity = count, mostly one feature per node
])
def test_bow_match_crowded_nodes(oracle, n_kf, n_f, cap, n_nodes, flip, seed):
    rng = np.random.default_rng(seed)
    ex = oa.ORBextractor(500, 1.2, 4, 20, 7, max_width=320, max_height=240, max_batch=2, keep_pyramid=False)
    pairs = []
    for p in range(2):
        desc_f = rng.integers(0, 256, (n_f, 32), dtype=np.uint8)
        # keyframe descriptors: copies of frame descriptors (with repeats, so that several keyframe features want
        # the same frame feature) with a few bits flipped
        src = rng.integers(0, max(1, n_f // 2), n_kf)
        desc_kf = desc_f[src].copy()
        for _ in range(flip):
            byte = rng.integers(0, 32, n_kf)
            desc_kf[np.arange(n_kf), byte] ^= (1 << rng.integers(0, 8, n_kf)).astype(np.uint8)
        f_side = made_up_side(rng, n_f, cap, n_nodes, desc_f)
        kf_side = made_up_side(rng, n_kf, cap, n_nodes, desc_kf)
        if p == 0 and n_nodes > 1:
            # give every copy its source's node, so that the twins can meet
            node_f = np.zeros(n_f, np.uint32)
            node_f[f_side[1][:f_side[2]]] = f_side[0][:f_side[2]]
            idx = kf_side[1][:kf_side[2]]
            nk = node_f[src[idx]]
            nk[nk == 0] = 1
            order = np.lexsort((idx, nk))
            kf_side[0][:kf_side[2]] = nk[order]
            kf_side[1][:kf_side[2]] = idx[order]
        valid = np.zeros(cap, np.uint8)
        valid[:n_kf] = rng.random(n_kf) < 0.9
        pairs.append((kf_side + (valid,), f_side))
    kf = tuple(np.stack([pairs[p][0][i] for p in range(2)]) for i in range(6))
    f = tuple(np.stack([pairs[p][1][i] for p in range(2)]) for i in range(5))
    valid2 = np.zeros((2, cap), np.uint8)
    valid2[:, :n_f] = rng.random((2, n_f)) < 0.9
    for check in (False, True):
        match, nm = run_device(ex, kf, f, 0.75, check)
        total = check_against_oracle(oracle, kf, f, [n_f, n_f], 0.75, check, match, nm)
        assert total > 10
        match12, nm12 = run_device_keyframes(ex, kf, f + (valid2,), 0.75, check)
        assert check_keyframes_against_oracle(oracle, kf, f + (valid2,), 0.75, check, match12, nm12) > 10
    ex.close()


def test_bow_match_arguments(oracle):
    ex = oa.ORBextractor(500, 1.2, 4, 20, 7, max_width=320, max_height=240, max_batch=1, keep_pyramid=False)
    with pytest.raises(oa.OrbxError):
        ex.search_by_bow_batch_device(1, 1, 1, 1, 1, 1, 1, 20000, 1, 1, 1, 1, 1, 100, 0.7, 1, 1, 1)   # capacity above 16384
    with pytest.raises(oa.OrbxError):
        ex.search_by_bow_batch_device(1, 0, 1, 1, 1, 1, 1, 100, 1, 1, 1, 1, 1, 100, 0.7, 1, 1, 1)     # null FeatureVector
    ex.close()


def test_bow_match_oversized_counts_are_cut_to_the_capacity(oracle):
    """FeatureVector counts above the arrays' capacity (a caller's mistake) are read as the capacity: the result equals
    the oracle's on full arrays, and pair 0's search does not run on into pair 1's slice."""
    rng = np.random.default_rng(9)
    cap, n_nodes = 256, 8
    ex = oa.ORBextractor(500, 1.2, 4, 20, 7, max_width=320, max_height=240, max_batch=2, keep_pyramid=False)
    sides = []
    for p in range(2):
        desc_f = rng.integers(0, 256, (cap, 32), dtype=np.uint8)
        desc_kf = desc_f[rng.integers(0, cap, cap)].copy()
        desc_kf[np.arange(cap), rng.integers(0, 32, cap)] ^= 1
        f_side, kf_side = list(made_up_side(rng, cap, cap, n_nodes, desc_f)), list(made_up_side(rng, cap, cap, n_nodes, desc_kf))
        for s in (f_side, kf_side):   # fill the arrays to the brim: every feature is in the FeatureVector
            node_of = rng.integers(1, n_nodes + 1, cap).astype(np.uint32)
            order = np.lexsort((np.arange(cap), node_of))
            s[0], s[1], s[2] = node_of[order], order.astype(np.uint32), np.int32(cap)
        sides.append((tuple(kf_side) + (np.ones(cap, np.uint8),), tuple(f_side)))
    kf = tuple(np.stack([sides[p][0][i] for p in range(2)]) for i in range(6))
    f = tuple(np.stack([sides[p][1][i] for p in range(2)]) for i in range(5))
    big_kf = kf[:2] + (np.array([cap + 1000, 2 ** 30], np.int32),) + kf[3:]
    big_f = f[:2] + (np.array([2 ** 31 - 1, cap + 1], np.int32),) + f[3:]
    match, nm = run_device(ex, big_kf, big_f, 0.75, True)
    assert check_against_oracle(oracle, kf, f, [cap, cap], 0.75, True, match, nm) > 10
    ex.close()
