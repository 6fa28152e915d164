"""GPU parity of the search inside ORBmatcher::Fuse(pKF, vpMapPoints, th) (reference src/ORBmatcher.cc:929-986) against the
CPU oracle, bit-exact (indices and integer distances), on the undistorted keypoints, grid and descriptors of real
extractions; and the second Fuse overload's / SearchBySim3's search (no chi-square test) through the plain window
search."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth
from hip_rt import DeviceArray

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_right,th", [(False, 3.0), (True, 3.0), (True, 6.0)])
def test_fuse_search_after_extraction(oracle, use_right, th):
    w, h, nf, nl, n = 640, 480, 1000, 8, 3
    imgs = synth.make_batch(n, w, h, 1, 55)
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    d_img = DeviceArray.from_host(imgs)
    d_kps, d_desc = DeviceArray((n, cap, 28), np.uint8, fill_byte=0), DeviceArray((n, cap, 32), np.uint8, fill_byte=0)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    d_un = DeviceArray((n, cap, 28), np.uint8, fill_byte=0)
    d_cs, d_ci = DeviceArray((n, 64 * 48 + 1), np.int32, fill_byte=0xFF), DeviceArray((n, cap), np.int32, fill_byte=0xFF)
    grid = (0.0, 0.0, float(np.float32(64) / np.float32(w)), float(np.float32(48) / np.float32(h)))
    ex.extract_batch_device(d_img.ptr, n, w, h, w, h * w, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=False)
    ex.undistort_grid_batch_device(n, d_kps.ptr, cap, d_n.ptr, np.array([500, 500, 320, 240], np.float32), np.zeros(5, np.float32),
                                   grid[0], grid[1], grid[2], grid[3], d_un.ptr, d_cs.ptr, d_ci.ptr, sync=True)
    counts = d_n.to_host()
    un = d_un.to_host().reshape(n, cap * 28).view(oa.KP_DTYPE).reshape(n, cap)
    desc, cs, ci = d_desc.to_host(), d_cs.to_host(), d_ci.to_host()
    rng = np.random.default_rng(int(th) + 10 * use_right)
    scale = ex.GetScaleFactors()
    inv_sigma2 = ex.GetInverseScaleSigmaSquares()
    u_right = np.where(rng.random((n, cap)) < 0.6, un["x"] - rng.uniform(0, 40, (n, cap)), -1.0).astype(np.float32)
    u_right[:, ::7] = 0.0                                   # mvuRight == 0 counts as a stereo measurement
    # the map points: the keyframe's own keypoints seen a little displaced (most inside the chi-square gate, some
    # outside), predicted on their own level or the next, plus windows that leave the grid and an empty one
    q_cap = cap + 8
    queries = np.zeros((n, q_cap), oa.WINDOW_QUERY_DTYPE)
    nq = np.zeros(n, np.int32)
    for f in range(n):
        k = int(counts[f])
        src = un[f, :k]
        q = queries[f]
        sig = (1.2 ** src["octave"]).astype(np.float32)
        q["u"][:k] = src["x"] + (rng.normal(0, 1.3, k) * sig).astype(np.float32)
        q["v"][:k] = src["y"] + (rng.normal(0, 1.3, k) * sig).astype(np.float32)
        level = np.clip(src["octave"] + rng.integers(0, 2, k), 0, nl - 1)
        q["min_level"][:k], q["max_level"][:k] = level - 1, level
        q["radius"][:k] = np.float32(th) * scale[level]
        q["ur"][:k] = np.where(u_right[f, :k] >= 0, u_right[f, :k] + (rng.normal(0, 1.3, k) * sig), q["u"][:k] - 10).astype(np.float32)
        noisy = desc[f, :k].copy()
        flip = rng.random((k, 32)) < 0.08
        noisy[flip] ^= (1 << rng.integers(0, 8, (k, 32)).astype(np.uint8))[flip]
        q["desc"][:k] = noisy
        extra = [(-500.0, 100.0, 30.0), (5000.0, 100.0, 30.0), (100.0, -900.0, 30.0), (100.0, 9000.0, 30.0), (320.0, 240.0, 0.0)]
        for e, (u, v, r) in enumerate(extra):
            q["u"][k + e], q["v"][k + e], q["radius"][k + e] = u, v, r
            q["min_level"][k + e], q["max_level"][k + e] = 0, 1
        nq[f] = k + len(extra)
    d_q = DeviceArray.from_host(queries.view(np.uint8).reshape(n, q_cap * 56))
    d_nq = DeviceArray.from_host(nq)
    d_ur = DeviceArray.from_host(u_right)
    d_bi = DeviceArray((n, q_cap), np.int32, fill_byte=0x55)
    d_bd = DeviceArray((n, q_cap), np.int32, fill_byte=0x55)
    ex.fuse_search_batch_device(n, d_un.ptr, cap, d_cs.ptr, d_ci.ptr, d_desc.ptr, d_ur.ptr if use_right else 0, grid[0], grid[1],
                                grid[2], grid[3], d_q.ptr, q_cap, d_nq.ptr, d_bi.ptr, d_bd.ptr, sync=True)
    bi, bd = d_bi.to_host(), d_bd.to_host()
    found = gated = 0
    for f in range(n):
        k = int(counts[f])
        xy = np.stack([un[f, :k]["x"], un[f, :k]["y"]], 1)
        for i in range(nq[f]):
            q = queries[f, i]
            args = (xy, un[f, :k]["octave"], desc[f, :k], u_right[f, :k] if use_right else None, cs[f], ci[f, :k], grid[0], grid[1],
                    grid[2], grid[3], float(q["u"]), float(q["v"]), float(q["ur"]), float(q["radius"]), int(q["max_level"]),
                    inv_sigma2, q["desc"])
            idx, dist = oracle.fuse_search(*args)
            assert (bi[f, i], bd[f, i]) == (idx, dist), (f, i)
            found += idx >= 0
            # how often the gate matters: the plain window search finds something else
            widx, _ = oracle.search_window(xy, un[f, :k]["octave"], desc[f, :k], None, None, cs[f], ci[f, :k], grid[0], grid[1], grid[2],
                                           grid[3], float(q["u"]), float(q["v"]), float(q["radius"]), int(q["min_level"]),
                                           int(q["max_level"]), 0.0, q["desc"])[:2]
            gated += widx != idx
    assert found > 0.5 * counts.sum() and gated > 20
    # the search of Fuse(pKF, Scw, ...) and SearchBySim3 (no chi-square test) is the plain window search
    ex.search_window_batch_device(n, d_un.ptr, cap, d_cs.ptr, d_ci.ptr, d_desc.ptr, 0, 0, grid[0], grid[1], grid[2], grid[3], d_q.ptr,
                                  q_cap, d_nq.ptr, d_bi.ptr, d_bd.ptr, sync=True)
    bi2, bd2 = d_bi.to_host(), d_bd.to_host()
    open_gate = np.zeros(nl, np.float32)                    # inverse sigmas of zero: nothing fails the chi-square test
    for f in range(n):
        k = int(counts[f])
        xy = np.stack([un[f, :k]["x"], un[f, :k]["y"]], 1)
        for i in range(0, nq[f], 3):
            q = queries[f, i]
            idx, dist = oracle.fuse_search(xy, un[f, :k]["octave"], desc[f, :k], None, cs[f], ci[f, :k], grid[0], grid[1], grid[2],
                                           grid[3], float(q["u"]), float(q["v"]), float(q["ur"]), float(q["radius"]),
                                           int(q["max_level"]), open_gate, q["desc"])
            assert (bi2[f, i], bd2[f, i]) == (idx, dist), (f, i)
    ex.close()
    for a in (d_img, d_kps, d_desc, d_n, d_un, d_cs, d_ci, d_q, d_nq, d_ur, d_bi, d_bd):
        a.free()
