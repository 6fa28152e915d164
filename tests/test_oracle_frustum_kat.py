"""Known-answer tests of the oracle's restatement of Frame::isInFrustum (reference src/Frame.cc:271-327),
MapPoint::PredictScale (src/MapPoint.cc:402-417) and the tests in front of the search of ORBmatcher::Fuse
(src/ORBmatcher.cc:881-927): cameras and points whose result can be read off the source."""
import numpy as np

import oracle_lib

IDENT = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], np.float32)
CAM = np.array([500, 500, 320, 240, 40], np.float32)     # fx, fy, cx, cy, mbf
BOUNDS = np.array([0, 640, 0, 480], np.float32)          # mnMinX, mnMaxX, mnMinY, mnMaxY
LOG12 = float(np.float32(np.log(np.float64(np.float32(1.2)))))
SCALE = np.array([np.float32(1.2) ** i for i in range(8)], np.float32)
POINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"), ("min_dist", "<f4"),
                  ("max_dist", "<f4"), ("flags", "<i4"), ("desc", "u1", (32,))])


def pt(x, y, z, normal=(0, 0, 1), min_dist=0.1, max_dist=100.0, flags=0):
    p = np.zeros(1, POINT)
    p["x"], p["y"], p["z"] = x, y, z
    p["nx"], p["ny"], p["nz"] = normal
    p["min_dist"], p["max_dist"], p["flags"] = min_dist, max_dist, flags
    return p


def frustum(p, T=IDENT, Ow=(0, 0, 0), limit=0.5):
    out, n = oracle_lib.is_in_frustum(T, Ow, CAM, BOUNDS, limit, LOG12, 8, p)
    return out[0], n


def fuse(p, T=IDENT, Ow=(0, 0, 0), th=3.0):
    u, v, ur, rad, lvl, ok = oracle_lib.fuse_project(T, Ow, CAM, BOUNDS, LOG12, 8, SCALE, th, p)
    return float(u[0]), float(v[0]), float(ur[0]), float(rad[0]), int(lvl[0]), bool(ok[0])


def test_log_is_the_correctly_rounded_float_logarithm():
    rng = np.random.default_rng(1)
    x = np.concatenate([
        np.array([1.0, 2.0, 0.5, 1.2, np.e, 1e-38, 1e-45, 3.4e38, 1.4142135, 1.4142137, 0.70710677, 0.9999999, 1.0000001], np.float32),
        np.exp(rng.uniform(-20, 20, 20000)).astype(np.float32),
        rng.uniform(0.5, 2.0, 20000).astype(np.float32),
        (np.float32(1.2) ** rng.integers(-8, 9, 2000)).astype(np.float32)])
    got = oracle_lib.log_pinned(x)
    want = np.log(x.astype(np.float64))
    assert np.abs(got - want).max() <= 4e-16 * np.maximum(1.0, np.abs(want)).max()
    assert (got.astype(np.float32) == want.astype(np.float32)).all()
    assert got[0] == 0.0


def test_point_on_the_axis_and_the_stored_projection():
    q, n = frustum(pt(0, 0, 2.0, max_dist=2.0))
    assert n == 1 and q["flags"] == 0
    assert (q["u"], q["v"], q["ur"], q["view_cos"], q["level"]) == (320.0, 240.0, 300.0, 1.0, 0)
    # one metre to the right at depth 4: u = 500 * 1 * 0.25 + 320
    q, n = frustum(pt(1.0, -0.5, 4.0, normal=(0, 0, 1), max_dist=8.0))
    assert (q["u"], q["v"], q["ur"]) == (445.0, 177.5, 435.0)
    # a translated camera: Pc = P + t, PO = P - Ow with Ow = -t
    T = IDENT.copy()
    T[:, 3] = (0.5, 0, 1.0)
    q, n = frustum(pt(0.5, 0, 1.0, max_dist=2.0), T=T, Ow=(-0.5, 0, -1.0))
    assert n == 1 and (q["u"], q["v"]) == (570.0, 240.0)
    # Observations() > 0 is handed on, a point marked "skip" stays skipped
    assert frustum(pt(0, 0, 2.0, flags=1))[0]["flags"] == 1
    assert frustum(pt(0, 0, 2.0, flags=3)) [0]["flags"] == 3 and frustum(pt(0, 0, 2.0, flags=2))[1] == 0


def test_depth_and_image_bounds():
    assert frustum(pt(0, 0, -1.0))[1] == 0                              # :285
    assert frustum(pt(0.64, 0, 1.0, normal=(0.5390, 0, 0.8423)))[0]["u"] == 640.0   # u == mnMaxX is inside (:293)
    assert frustum(pt(0.6402, 0, 1.0, normal=(0.5390, 0, 0.8423)))[1] == 0
    assert frustum(pt(-0.64, 0, 1.0, normal=(-0.5390, 0, 0.8423)))[0]["u"] == 0.0
    assert frustum(pt(0, 0.4802, 1.0, normal=(0, 0.4329, 0.9014)))[1] == 0
    # Fuse uses KeyFrame::IsInImage: the upper bounds are excluded
    assert fuse(pt(0.64, 0, 1.0, normal=(0.5390, 0, 0.8423)))[5] is False
    assert fuse(pt(0.6398, 0, 1.0, normal=(0.5390, 0, 0.8423)))[5] is True
    assert fuse(pt(-0.64, 0, 1.0, normal=(-0.5390, 0, 0.8423)))[5] is True
    assert fuse(pt(0, 0, -1.0))[5] is False


def test_distance_invariance_region():
    # 0.8 * mfMinDistance <= dist <= 1.2 * mfMaxDistance (:304, MapPoint.cc:373-383); dist = 2
    assert frustum(pt(0, 0, 2.0, min_dist=2.5, max_dist=10))[1] == 1    # 0.8 * 2.5 = 2: not below
    assert frustum(pt(0, 0, 2.0, min_dist=2.51, max_dist=10))[1] == 0
    assert frustum(pt(0, 0, 3.0, min_dist=0.1, max_dist=2.5))[1] == 1   # 1.2 * 2.5 = 3: not above
    assert frustum(pt(0, 0, 3.0, min_dist=0.1, max_dist=2.49))[1] == 0
    assert fuse(pt(0, 0, 2.0, min_dist=2.51, max_dist=10))[5] is False
    assert fuse(pt(0, 0, 3.0, min_dist=0.1, max_dist=2.49))[5] is False


def test_viewing_angle():
    # PO = (0, 0, 2); a normal at 60 degrees gives exactly 0.5, which is not below the limit (:312)
    q, n = frustum(pt(0, 0, 2.0, normal=(0, 0.8660254, 0.5)))
    assert n == 1 and q["view_cos"] == 0.5
    assert frustum(pt(0, 0, 2.0, normal=(0, 0.8660254, 0.4999)))[1] == 0
    assert frustum(pt(0, 0, 2.0, normal=(0, 0.8660254, 0.4999)), limit=0.4)[1] == 1
    assert frustum(pt(0, 0, 2.0, normal=(0, 0, -1)))[1] == 0
    # Fuse: PO . Pn >= 0.5 * dist (:921)
    assert fuse(pt(0, 0, 2.0, normal=(0, 0.8660254, 0.5)))[5] is True
    assert fuse(pt(0, 0, 2.0, normal=(0, 0.8660254, 0.4999)))[5] is False


def test_predicted_level():
    # ceil(log(mfMaxDistance / dist) / log(1.2)), clamped to 0 .. 7; dist = 2
    for max_dist, level in ((2.0, 0), (2.6, 2), (1.8, 0), (2.0 * 1.2 ** 2.5, 3), (2.0 * 1.2 ** 6.5, 7), (200.0, 7)):
        min_dist = 0.01
        if 1.2 * max_dist < 2.0:
            continue
        assert frustum(pt(0, 0, 2.0, min_dist=min_dist, max_dist=max_dist))[0]["level"] == level, max_dist
        u, v, ur, rad, lvl, ok = fuse(pt(0, 0, 2.0, min_dist=min_dist, max_dist=max_dist), th=3.0)
        assert ok and lvl == level and rad == float(np.float32(3.0) * SCALE[level])   # :927
    # a point nearer than its maximum distance by less than one level still rounds up
    assert frustum(pt(0, 0, 2.0, max_dist=2.01))[0]["level"] == 1
    # the Fuse record: u, v, ur as :892-899
    u, v, ur, rad, lvl, ok = fuse(pt(1.0, -0.5, 4.0, max_dist=4.2))
    assert (u, v, ur) == (445.0, 177.5, 435.0)


def reloc(p, T=IDENT, Ow=(0, 0, 0), th=10.0):
    out, n = oracle_lib.project_guided(T, Ow, CAM, BOUNDS, LOG12, 8, SCALE, th, p, None, True)
    return out[0], n


def test_relocalisation_projection_has_no_depth_and_no_viewing_test():
    # src/ORBmatcher.cc:1534-1563: bounds closed, distance region, PredictScale, radius = th * scale[level]
    q, n = reloc(pt(1.0, -0.5, 4.0, max_dist=8.0))
    assert n == 1 and (q["u"], q["v"], q["flags"], q["level"]) == (445.0, 177.5, 1, 4)   # ratio 2 / ... dist = 4.153
    assert q["radius"] == np.float32(10.0) * SCALE[4]
    # a point behind the camera projects through the centre and is kept (there is no "PcZ < 0" here) ...
    q, n = reloc(pt(0.2, 0.1, -2.0, max_dist=4.0))
    assert n == 1 and (q["u"], q["v"]) == (270.0, 215.0)
    # ... and so is one seen from behind its normal
    assert reloc(pt(0, 0, 2.0, normal=(0, 0, -1), max_dist=2.0))[1] == 1
    # the loop closer's projection (:321-360) drops both
    loop = lambda p: oracle_lib.project_guided(IDENT, (0, 0, 0), CAM, BOUNDS, LOG12, 8, SCALE, 10.0, p, None, False)
    assert loop(pt(0.2, 0.1, -2.0, max_dist=4.0))[1] == 0 and loop(pt(0, 0, 2.0, normal=(0, 0, -1), max_dist=2.0))[1] == 0
    out, n = loop(pt(1.0, -0.5, 4.0, max_dist=8.0))
    assert n == 1 and (out[0]["u"], out[0]["v"], out[0]["flags"], out[0]["level"]) == (445.0, 177.5, 1, 4)
    # u == mnMaxX: inside for the Frame's bounds, outside for KeyFrame::IsInImage
    edge = pt(0.64, 0, 1.0, normal=(0.5390, 0, 0.8423), max_dist=2.0)
    assert reloc(edge)[1] == 1 and loop(edge)[1] == 0
    # outside the distance region, skipped on input
    assert reloc(pt(0, 0, 3.0, min_dist=0.1, max_dist=2.49))[1] == 0
    q, n = reloc(pt(0, 0, 2.0, flags=2))
    assert n == 0 and (q["flags"], q["radius"]) == (3, -1.0)


def test_sim3_projection():
    # src/ORBmatcher.cc:1195-1226: camera 1 at the origin, camera 2 = camera 1 scaled by 1/2 and shifted by (0.1, 0, 0)
    S21 = np.array([[0.5, 0, 0, 0.1], [0, 0.5, 0, 0], [0, 0, 0.5, 0]], np.float32)
    sim3 = lambda p, T=IDENT, S=S21, th=7.5: [a[0] for a in oracle_lib.sim3_project(T, S, CAM, BOUNDS, LOG12, 8, SCALE, th, p)]
    # P = (1, -0.5, 4) -> camera 2: (0.6, -0.25, 2); u = 500 * 0.3 + 320, v = 500 * -0.125 + 240
    u, v, rad, lvl, ok = sim3(pt(1.0, -0.5, 4.0, max_dist=4.0))
    assert ok and (u, v) == (470.0, 177.5)
    # the distance is the length of the camera-2 point, 2.103: ratio 1.902 -> level 4
    assert lvl == 4 and rad == np.float32(7.5) * SCALE[4]
    # depth in camera 2 decides, the normal does not; IsInImage's half-open bounds
    assert not sim3(pt(0, 0, -1.0))[4]
    assert sim3(pt(0, 0, 2.0, normal=(0, 0, -1), max_dist=1.0))[4]
    assert not sim3(pt(1.08, 0, 2.0, max_dist=2.0))[4]            # x2 = 0.64, z2 = 1: u = 640
    assert sim3(pt(1.0799, 0, 2.0, max_dist=2.0))[4]
    # the distance region is tested on the camera-2 length: 1 for (0, 0, 2)
    assert sim3(pt(0, 0, 2.0, min_dist=1.25, max_dist=5.0))[4] and not sim3(pt(0, 0, 2.0, min_dist=1.26, max_dist=5.0))[4]
    assert not sim3(pt(0, 0, 2.0, max_dist=1.0, flags=2))[4]
