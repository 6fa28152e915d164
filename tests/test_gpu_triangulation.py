"""GPU parity of ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:686-852) against the CPU oracle,
bit-exact: vMatches12 and nmatches, on the chain extract -> undistort -> bag of words -> match between shifted views
of one scene (with a fundamental matrix that fits the shift and others that do not), and on made-up FeatureVectors
with crowded nodes."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth
from hip_rt import DeviceArray

pytestmark = pytest.mark.gpu


def run_device(ex, k1, k2, f12, epipole, only_stereo, check):
    """k = (fv_node, fv_idx, fv_n, desc, kps_un, has_mp, u_right or None) with a leading pair axis."""
    n, cap_1, cap_2 = k1[0].shape[0], k1[0].shape[1], k2[0].shape[1]

    def up(side):
        return [None if a is None else DeviceArray.from_host(np.ascontiguousarray(a.view(np.uint8) if a.dtype == oa.KP_DTYPE else a))
                for a in side]
    d1, d2 = up(k1), up(k2)
    d_f, d_e = DeviceArray.from_host(np.ascontiguousarray(f12, np.float32)), DeviceArray.from_host(np.ascontiguousarray(epipole, np.float32))
    d_match = DeviceArray((n, cap_1), np.int32, fill_byte=0x55)
    d_nm = DeviceArray((n,), np.int32, fill_byte=0x55)
    p = lambda a: 0 if a is None else a.ptr  # noqa: E731
    ex.search_for_triangulation_batch_device(n, p(d1[0]), p(d1[1]), p(d1[2]), p(d1[3]), p(d1[4]), p(d1[5]), p(d1[6]), cap_1,
                                             p(d2[0]), p(d2[1]), p(d2[2]), p(d2[3]), p(d2[4]), p(d2[5]), p(d2[6]), cap_2,
                                             d_f.ptr, d_e.ptr, only_stereo, check, d_match.ptr, d_nm.ptr, sync=True)
    out = d_match.to_host(), d_nm.to_host()
    for a in d1 + d2 + [d_f, d_e, d_match, d_nm]:
        if a is not None:
            a.free()
    return out


def check_against_oracle(oracle, ex, k1, k2, counts_1, counts_2, f12, epipole, only_stereo, check, match, nm):
    scale, sigma2 = ex.GetScaleFactors(), ex.GetScaleSigmaSquares()
    total = 0
    for p in range(k1[0].shape[0]):
        def side(k, cnt):
            nf, c = int(k[2][p]), int(cnt[p])
            kp = k[4][p, :c]
            return (k[0][p, :nf], k[1][p, :nf], k[3][p, :c], np.stack([kp["x"], kp["y"]], 1), kp["angle"], kp["octave"], k[5][p, :c],
                    None if k[6] is None else k[6][p, :c])
        exp_n, exp_m = oracle.search_for_triangulation(side(k1, counts_1), side(k2, counts_2), f12[p], epipole[p], scale, sigma2,
                                                       only_stereo, check)
        assert nm[p] == exp_n, "nmatches of pair %d: %d, expected %d" % (p, nm[p], exp_n)
        assert (match[p, :counts_1[p]] == exp_m).all(), "matches of pair %d" % p
        assert (match[p, counts_1[p]:] == -1).all()
        total += exp_n
    return total


@pytest.mark.parametrize("only_stereo,check,use_right,seed", [(False, True, False, 1), (False, False, True, 2), (True, True, True, 3)])
def test_triangulation_match_after_extraction(oracle, only_stereo, check, use_right, seed):
    """Four views of one scene, each shifted 6 px along x from the one before (epipolar lines = image rows); pair p
    matches view p with view p + 1.  Pairs 0 and 1 get the matrix of that motion, pair 2 a slightly rotated one, pair 3
    an epipole inside the image."""
    from test_gpu_bow import run_bow
    w, h, nf, n = 640, 480, 1000, 4
    base = synth.make_frame(w + 32, h + 8, 2, 17 + seed)
    frames = np.stack([np.ascontiguousarray(base[4:4 + h, 6 * i:6 * i + w]) for i in range(n)])
    ex = oa.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    d_img = DeviceArray.from_host(frames)
    d_kps, d_desc = DeviceArray((n, cap, 28), np.uint8, fill_byte=0), DeviceArray((n, cap, 32), np.uint8, fill_byte=0)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    d_un = DeviceArray((n, cap, 28), np.uint8, fill_byte=0)
    d_cs, d_ci = DeviceArray((n, 64 * 48 + 1), np.int32, fill_byte=0xFF), DeviceArray((n, cap), np.int32, fill_byte=0xFF)
    grid = (0.0, 0.0, float(np.float32(64) / np.float32(w)), float(np.float32(48) / np.float32(h)))
    ex.extract_batch_device(d_img.ptr, n, w, h, w, h * w, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=False)
    ex.undistort_grid_batch_device(n, d_kps.ptr, cap, d_n.ptr, np.array([500, 500, 320, 240], np.float32),
                                   np.array([0.05, -0.02, 0.001, 0.001, 0.0], np.float32), grid[0], grid[1], grid[2], grid[3],
                                   d_un.ptr, d_cs.ptr, d_ci.ptr, sync=True)
    counts, desc = d_n.to_host(), d_desc.to_host()
    un = d_un.to_host().reshape(n, cap * 28).view(oa.KP_DTYPE).reshape(n, cap)
    ovoc = oracle.random_vocabulary(10, 3, seed)
    voc = oa.ORBVocabulary(10, 3, 0, 0, ovoc.parent, ovoc.is_leaf, ovoc.desc, ovoc.weight)
    out = run_bow(ex, voc, d_desc, cap, d_n, n, 2)
    fnode, fidx, fn = out[6], out[7], out[8]
    rng = np.random.default_rng(seed)
    has_mp = (rng.random((n, cap)) < 0.3).astype(np.uint8)
    u_right = np.where(rng.random((n, cap)) < 0.5, un["x"] - rng.uniform(2, 40, (n, cap)), -1.0).astype(np.float32) if use_right else None
    nxt = np.roll(np.arange(n), -1)
    k1 = (fnode, fidx, fn, desc, un, has_mp, u_right)
    k2 = (fnode[nxt], fidx[nxt], fn[nxt], desc[nxt], un[nxt], has_mp[nxt], None if u_right is None else u_right[nxt])
    f12 = np.zeros((n, 3, 3), np.float32)
    f12[:] = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    th = 0.004
    f12[2] = np.array([[0, 0, np.sin(th)], [0, 0, -np.cos(th)], [-np.sin(th) * 320, np.cos(th), 0]], np.float32)
    epipole = np.array([[-1e5, 240], [1e5, 240], [-3000, 200], [300, 240]], np.float32)
    match, nm = run_device(ex, k1, k2, f12, epipole, only_stereo, check)
    total = check_against_oracle(oracle, ex, k1, k2, counts, counts[nxt], f12, epipole, only_stereo, check, match, nm)
    assert total > (20 if only_stereo else 100)
    voc.close()
    ex.close()
    for a in (d_img, d_kps, d_desc, d_n, d_un, d_cs, d_ci):
        a.free()


@pytest.mark.parametrize("n_1,n_2,cap,n_nodes,seed", [(1800, 2000, 2048, 40, 1), (3000, 3500, 4096, 3, 2), (700, 16000, 16384, 1, 3),
                                                      (64, 64, 64, 64, 4)])
def test_triangulation_match_crowded_nodes(oracle, n_1, n_2, cap, n_nodes, seed):
    """Made-up FeatureVectors with up to 16000 candidates per node, descriptors in clusters (many equal distances: the
    last of equals must win), positions near and off the epipolar rows, an epipole inside the image."""
    from test_gpu_bowmatch import made_up_side
    rng = np.random.default_rng(seed)
    ex = oa.ORBextractor(500, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=2, keep_pyramid=False)
    sides = []
    base = rng.integers(0, 256, (6, 32), dtype=np.uint8)
    for count in (n_1, n_2):
        per_pair = []
        for p in range(2):
            d = base[rng.integers(0, 6, count)].copy()
            flip = rng.random((count, 32)) < 0.03
            d[flip] ^= (1 << rng.integers(0, 8, (count, 32)).astype(np.uint8))[flip]
            node, idx, fv_n, dd, kps = made_up_side(rng, count, cap, n_nodes, d)
            kps["x"][:count] = rng.uniform(20, 620, count)
            kps["y"][:count] = rng.choice(np.array([100.0, 100.5, 101.0, 102.0, 103.0, 250.0], np.float32), count)
            kps["octave"][:count] = rng.integers(0, 8, count)
            mp = np.zeros(cap, np.uint8)
            mp[:count] = rng.random(count) < 0.2
            ur = np.full(cap, -1.0, np.float32)
            ur[:count] = np.where(rng.random(count) < 0.4, kps["x"][:count] - 10, -1.0)
            per_pair.append((node, idx, fv_n, dd, kps, mp, ur))
        sides.append(tuple(np.stack([per_pair[p][i] for p in range(2)]) for i in range(7)))
    k1, k2 = sides
    f12 = np.zeros((2, 3, 3), np.float32)
    f12[:] = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    epipole = np.array([[320, 101], [-1e5, 100]], np.float32)
    for only_stereo, check in ((False, True), (True, False)):
        match, nm = run_device(ex, k1, k2, f12, epipole, only_stereo, check)
        total = check_against_oracle(oracle, ex, k1, k2, [n_1, n_1], [n_2, n_2], f12, epipole, only_stereo, check, match, nm)
        assert total > (5 if n_1 > 100 else 0)
    # no stereo measurements at all: null mvuRight arrays
    k1n, k2n = k1[:6] + (None,), k2[:6] + (None,)
    match, nm = run_device(ex, k1n, k2n, f12, epipole, False, True)
    check_against_oracle(oracle, ex, k1n, k2n, [n_1, n_1], [n_2, n_2], f12, epipole, False, True, match, nm)
    with pytest.raises(oa.OrbxError):
        ex.search_for_triangulation_batch_device(1, 16, 16, 16, 16, 16, 16, 0, 20000, 16, 16, 16, 16, 16, 16, 0, 100, 16, 16, 0, 1, 16, 16)
    with pytest.raises(oa.OrbxError):
        ex.search_for_triangulation_batch_device(1, 16, 16, 16, 16, 16, 16, 0, 100, 16, 16, 16, 16, 16, 16, 0, 100, 0, 16, 0, 1, 16, 16)
    ex.close()
