"""GPU parity of the windowed best-match search (Frame::GetFeaturesInArea, reference
src/Frame.cc:338-396, + the candidate loop of ORBmatcher::SearchByProjection, reference
src/ORBmatcher.cc:1427-1455) against the CPU oracle, bit-exact (indices and integer distances), on
the undistorted keypoints, grid and descriptors of real extractions."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth
from hip_rt import DeviceArray

pytestmark = pytest.mark.gpu

K = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)            # TUM1.yaml
DIST = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)


@pytest.mark.parametrize("use_right,use_taken", [(False, False), (True, True)])
def test_window_search_after_extraction(oracle, use_right, use_taken):
    w, h, nf, nl, n = 640, 480, 1000, 8, 2
    imgs = synth.make_batch(n, w, h, 1, 90)
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    d_img = DeviceArray.from_host(imgs)
    d_kps, d_desc = DeviceArray((n, cap, 28), np.uint8, fill_byte=0), DeviceArray((n, cap, 32), np.uint8)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    d_un = DeviceArray((n, cap, 28), np.uint8, fill_byte=0)
    d_cs = DeviceArray((n, 64 * 48 + 1), np.int32, fill_byte=0xFF)
    d_ci = DeviceArray((n, cap), np.int32, fill_byte=0xFF)
    c = oracle.undistort_points(np.array([[0, 0], [w, 0], [0, h], [w, h]], np.float32), K, DIST)
    min_x, max_x = min(c[0, 0], c[2, 0]), max(c[1, 0], c[3, 0])
    min_y, max_y = min(c[0, 1], c[1, 1]), max(c[2, 1], c[3, 1])
    inv_w = np.float32(64) / np.float32(max_x - min_x)
    inv_h = np.float32(48) / np.float32(max_y - min_y)
    ex.extract_batch_device(d_img.ptr, n, w, h, w, h * w, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=False)
    ex.undistort_grid_batch_device(n, d_kps.ptr, cap, d_n.ptr, K, DIST, float(min_x), float(min_y), float(inv_w),
                                   float(inv_h), d_un.ptr, d_cs.ptr, d_ci.ptr, sync=True)
    counts = d_n.to_host()
    un = d_un.to_host().reshape(n, cap * 28).view(oa.KP_DTYPE).reshape(n, cap)
    desc, cs, ci = d_desc.to_host(), d_cs.to_host(), d_ci.to_host()
    rng = np.random.default_rng(17)
    # queries: keypoints of the OTHER frame moved by a few pixels (so windows hold several candidates and
    # descriptors are plausible), plus windows that leave the grid, empty ones, and every level-range form
    q_cap = 1300
    queries = np.zeros((n, q_cap), oa.WINDOW_QUERY_DTYPE)
    nq = np.zeros(n, np.int32)
    for f in range(n):
        o = 1 - f
        k = min(counts[o], q_cap - 40)
        src = un[o, :k]
        q = queries[f]
        q["u"][:k] = src["x"] + rng.uniform(-6, 6, k).astype(np.float32)
        q["v"][:k] = src["y"] + rng.uniform(-6, 6, k).astype(np.float32)
        q["radius"][:k] = (rng.choice([7.0, 15.0, 25.0], k) * 1.2 ** src["octave"]).astype(np.float32)
        form = rng.integers(0, 4, k)
        q["min_level"][:k] = np.where(form == 0, -1, np.where(form == 1, src["octave"], np.where(form == 2, 0, src["octave"] - 1)))
        q["max_level"][:k] = np.where(form == 0, -1, np.where(form == 1, -1, np.where(form == 2, src["octave"], src["octave"] + 1)))
        q["ur"][:k] = q["u"][:k] - rng.uniform(0, 40, k).astype(np.float32)
        q["desc"][:k] = desc[o, :k]
        flip = rng.random((k, 32)) < 0.03
        q["desc"][:k] ^= (flip * rng.integers(1, 256, (k, 32))).astype(np.uint8)
        extra = [(-500.0, 100.0, 30.0), (5000.0, 100.0, 30.0), (100.0, -900.0, 30.0), (100.0, 9000.0, 30.0),
                 (float(min_x) + 2, float(min_y) + 2, 60.0), (float(max_x) - 2, float(max_y) - 2, 60.0),
                 (320.0, 240.0, 400.0), (320.0, 240.0, 0.0)]
        for e, (u, v, r) in enumerate(extra):
            q["u"][k + e], q["v"][k + e], q["radius"][k + e] = u, v, r
            q["min_level"][k + e], q["max_level"][k + e] = -1, -1
            q["desc"][k + e] = rng.integers(0, 256, 32)
        nq[f] = k + len(extra)
    u_right = np.where(rng.random((n, cap)) < 0.6, rng.uniform(0, 640, (n, cap)), -1).astype(np.float32)
    taken = (rng.random((n, cap)) < 0.3).astype(np.uint8)
    d_q = DeviceArray.from_host(queries.view(np.uint8).reshape(n, q_cap * 56))
    d_nq = DeviceArray.from_host(nq)
    d_ur, d_tk = DeviceArray.from_host(u_right), DeviceArray.from_host(taken)
    d_bi = DeviceArray((n, q_cap), np.int32, fill_byte=0x55)
    d_bd = DeviceArray((n, q_cap), np.int32, fill_byte=0x55)
    ex.search_window_batch_device(n, d_un.ptr, cap, d_cs.ptr, d_ci.ptr, d_desc.ptr, d_ur.ptr if use_right else 0,
                                  d_tk.ptr if use_taken else 0, float(min_x), float(min_y), float(inv_w), float(inv_h),
                                  d_q.ptr, q_cap, d_nq.ptr, d_bi.ptr, d_bd.ptr, sync=True)
    bi, bd = d_bi.to_host(), d_bd.to_host()
    hits = 0
    for f in range(n):
        k = counts[f]
        xy = np.stack([un[f, :k]["x"], un[f, :k]["y"]], 1)
        octv = un[f, :k]["octave"].astype(np.int32)
        for j in range(nq[f]):
            q = queries[f, j]
            exp = oracle.search_window(xy, octv, desc[f, :k], u_right[f, :k] if use_right else None,
                                       taken[f, :k] if use_taken else None, cs[f], ci[f], min_x, min_y, inv_w, inv_h,
                                       q["u"], q["v"], q["radius"], int(q["min_level"]), int(q["max_level"]), q["ur"], q["desc"])
            assert (bi[f, j], bd[f, j]) == exp, "frame %d query %d" % (f, j)
            hits += exp[0] >= 0
        assert (bi[f, nq[f]:] == 0x55555555).all()      # nothing is written past the query count
    assert hits > 0.5 * nq.sum()
    ex.close()
    for a in (d_img, d_kps, d_desc, d_n, d_un, d_cs, d_ci, d_q, d_nq, d_ur, d_tk, d_bi, d_bd):
        a.free()
