"""GPU parity of the map-point projections -- Frame::isInFrustum with MapPoint::PredictScale (reference
src/Frame.cc:271-327, src/MapPoint.cc:402-417) and the tests in front of the search of ORBmatcher::Fuse
(src/ORBmatcher.cc:881-927) -- against the CPU oracle, bit for bit in every field, and the two device chains they
feed: projection -> local-map search, projection -> Fuse search; and the projections of the relocalisation's and the
loop closer's SearchByProjection (src/ORBmatcher.cc:1534-1563, :321-360) -> guided search; SearchBySim3's projection
(:1195-1226) -> window search."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth
from hip_rt import DeviceArray

pytestmark = pytest.mark.gpu

W, H, NL = 640, 480, 8
CAM = np.array([517.3, 516.5, 318.6, 255.3, 40.0], np.float32)
BOUNDS = np.array([0.0, W, 0.0, H], np.float32)


def _pose(rng, big=False):
    """A rotation by a small (or any) angle about a random axis and a translation, in float32 as a cv::Mat holds it."""
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = rng.uniform(-3.1, 3.1) if big else rng.uniform(-0.15, 0.15)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    t = rng.uniform(-0.5, 0.5, 3)
    T = np.concatenate([R, t[:, None]], 1).astype(np.float32)
    Ow = (-(T[:, :3].astype(np.float64).T @ T[:, 3].astype(np.float64))).astype(np.float32)
    return T, Ow


def _params(limit=0.5, th=3.0):
    p = oa.FrustumParams()
    p.fx, p.fy, p.cx, p.cy, p.mbf = [float(v) for v in CAM]
    p.min_x, p.max_x, p.min_y, p.max_y = [float(v) for v in BOUNDS]
    p.view_cos_limit, p.th = limit, th
    p.log_scale_factor = float(np.float32(np.log(np.float64(np.float32(1.2)))))
    return p


def _points_from_keypoints(rng, T, Ow, un, desc, nl, spread=1.0):
    """Map points that project onto the keypoints `un` (with a little noise): depth, normal, distance range and descriptor
    as a map would hold them."""
    k = len(un)
    z = rng.uniform(0.8, 12.0, k)
    x = (un["x"] + rng.normal(0, 0.7 * spread, k) - CAM[2]) * z / CAM[0]
    y = (un["y"] + rng.normal(0, 0.7 * spread, k) - CAM[3]) * z / CAM[1]
    Pc = np.stack([x, y, z], 1)
    R, t = T[:, :3].astype(np.float64), T[:, 3].astype(np.float64)
    Pw = (Pc - t) @ R                                            # R^T (Pc - t)
    po = Pw - Ow.astype(np.float64)
    dist = np.linalg.norm(po, axis=1)
    normal = po / dist[:, None] + rng.normal(0, 0.35, (k, 3))   # some beyond 60 degrees
    normal /= np.linalg.norm(normal, axis=1)[:, None]
    pts = np.zeros(k, oa.FRUSTUM_POINT_DTYPE)
    pts["x"], pts["y"], pts["z"] = Pw[:, 0], Pw[:, 1], Pw[:, 2]
    pts["nx"], pts["ny"], pts["nz"] = normal[:, 0], normal[:, 1], normal[:, 2]
    pts["max_dist"] = dist * 1.2 ** (un["octave"] + rng.uniform(-1.2, 0.4, k))   # a few outside the invariance region
    pts["min_dist"] = pts["max_dist"] / np.float32(1.2 ** (nl - 1))
    pts["flags"] = rng.choice(np.array([1, 1, 0, 2, 3], np.int32), k, p=[0.5, 0.2, 0.2, 0.05, 0.05])
    noisy = desc.copy()
    flip = rng.random((k, 32)) < 0.06
    noisy[flip] ^= (1 << rng.integers(0, 8, (k, 32)).astype(np.uint8))[flip]
    pts["desc"] = noisy
    return pts


def _edge_points(rng, T, Ow, m):
    """Points all over the place: behind the camera, outside the image, on the optical centre, degenerate distances."""
    pts = np.zeros(m, oa.FRUSTUM_POINT_DTYPE)
    P = rng.uniform(-15, 15, (m, 3))
    pts["x"], pts["y"], pts["z"] = P[:, 0], P[:, 1], P[:, 2]
    nrm = rng.normal(size=(m, 3))
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    pts["nx"], pts["ny"], pts["nz"] = nrm[:, 0], nrm[:, 1], nrm[:, 2]
    pts["max_dist"] = np.exp(rng.uniform(-2, 5, m))
    pts["min_dist"] = pts["max_dist"] * rng.uniform(0.0, 0.5, m)
    pts["flags"] = rng.integers(0, 4, m)
    pts["desc"] = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    # the camera centre itself (dist 0: the ratio is infinite), a zero and a negative maximum distance, zero depth
    pts[0]["x"], pts[0]["y"], pts[0]["z"] = Ow
    pts[0]["min_dist"], pts[0]["flags"] = 0.0, 0
    pts[1]["max_dist"], pts[1]["min_dist"], pts[1]["flags"] = 0.0, 0.0, 0
    pts[2]["max_dist"], pts[2]["min_dist"], pts[2]["flags"] = -1.0, -5.0, 0
    return pts


def _extract(n, seed):
    imgs = synth.make_batch(n, W, H, 1, seed)
    ex = oa.ORBextractor(1000, 1.2, NL, 20, 7, max_width=W, max_height=H, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(W, H)
    d = {"img": DeviceArray.from_host(imgs), "kps": DeviceArray((n, cap, 28), np.uint8, fill_byte=0),
         "desc": DeviceArray((n, cap, 32), np.uint8, fill_byte=0), "n": DeviceArray((n,), np.int32, fill_byte=0),
         "un": DeviceArray((n, cap, 28), np.uint8, fill_byte=0), "cs": DeviceArray((n, 64 * 48 + 1), np.int32, fill_byte=0xFF),
         "ci": DeviceArray((n, cap), np.int32, fill_byte=0xFF)}
    grid = (0.0, 0.0, float(np.float32(64) / np.float32(W)), float(np.float32(48) / np.float32(H)))
    ex.extract_batch_device(d["img"].ptr, n, W, H, W, H * W, d["kps"].ptr, cap, d["desc"].ptr, d["n"].ptr, sync=False)
    ex.undistort_grid_batch_device(n, d["kps"].ptr, cap, d["n"].ptr, CAM[:4].copy(), np.zeros(5, np.float32), grid[0], grid[1],
                                   grid[2], grid[3], d["un"].ptr, d["cs"].ptr, d["ci"].ptr, sync=True)
    return ex, cap, d, grid


def _same_bits(a, b, fields):
    for name in fields:
        x, y = np.ascontiguousarray(a[name]), np.ascontiguousarray(b[name])
        if x.dtype.kind == "f":
            x, y = x.view(np.uint32), y.view(np.uint32)
        if not (x == y).all():
            return name
    return None


@pytest.mark.parametrize("limit,th", [(0.5, 1.0), (0.3, 3.0)])
def test_is_in_frustum_then_local_map_search(oracle, limit, th):
    n = 3
    ex, cap, d, grid = _extract(n, 91)
    rng = np.random.default_rng(int(th * 10))
    counts = d["n"].to_host()
    un = d["un"].to_host().reshape(n, cap * 28).view(oa.KP_DTYPE).reshape(n, cap)
    desc, cs, ci = d["desc"].to_host(), d["cs"].to_host(), d["ci"].to_host()
    extra = 300
    q_cap = cap + extra + 5
    points = np.zeros((n, q_cap), oa.FRUSTUM_POINT_DTYPE)
    nq, poses, centres = np.zeros(n, np.int32), np.zeros((n, 12), np.float32), np.zeros((n, 3), np.float32)
    for f in range(n):
        T, Ow = _pose(rng)
        poses[f], centres[f] = T.ravel(), Ow
        k = int(counts[f])
        pts = np.concatenate([_points_from_keypoints(rng, T, Ow, un[f, :k], desc[f, :k], NL), _edge_points(rng, T, Ow, extra)])
        pts = pts[rng.permutation(len(pts))]
        nq[f] = len(pts) if f else len(pts) - 17        # the tail of frame 0 is not part of the list
        points[f, :len(pts)] = pts
    d_pts = DeviceArray.from_host(points.view(np.uint8).reshape(n, q_cap * 68))
    d_nq, d_pose, d_ow = DeviceArray.from_host(nq), DeviceArray.from_host(poses), DeviceArray.from_host(centres)
    d_q = DeviceArray((n, q_cap, 56), np.uint8, fill_byte=0x77)
    d_nv = DeviceArray((n,), np.int32, fill_byte=0x55)
    prm = _params(limit, th)
    ex.is_in_frustum_batch_device(n, prm, d_pose.ptr, d_ow.ptr, d_pts.ptr, q_cap, d_nq.ptr, d_q.ptr, d_nv.ptr, sync=True)
    got = d_q.to_host().reshape(n, q_cap * 56).view(oa.LOCAL_QUERY_DTYPE).reshape(n, q_cap)
    nv = d_nv.to_host()
    want = []
    for f in range(n):
        w, n_view = oracle.is_in_frustum(poses[f], centres[f], CAM, BOUNDS, limit, prm.log_scale_factor, NL, points[f, :nq[f]])
        want.append(w)
        assert nv[f] == n_view and n_view > 0.3 * counts[f], (f, nv[f], n_view)
        assert _same_bits(got[f, :nq[f]], w, ("u", "v", "ur", "level", "view_cos", "flags", "desc")) is None, f
        assert (d_q.to_host().reshape(n, q_cap, 56)[f, nq[f]:] == 0x77).all()   # nothing written past the list
        assert len(set(w["level"][w["flags"] < 2].tolist())) >= 5
    # the records go straight into the local-map search
    d_match, d_qm = DeviceArray((n, cap), np.int32, fill_byte=0x55), DeviceArray((n, q_cap), np.int32, fill_byte=0x55)
    d_nm = DeviceArray((n,), np.int32, fill_byte=0x55)
    ex.search_local_map_batch_device(n, d["un"].ptr, cap, d["cs"].ptr, d["ci"].ptr, d["desc"].ptr, 0, 0, grid[0], grid[1], grid[2],
                                     grid[3], th, 0.8, d_q.ptr, q_cap, d_nq.ptr, d_match.ptr, d_qm.ptr, d_nm.ptr, sync=True)
    match, qm, nm = d_match.to_host(), d_qm.to_host(), d_nm.to_host()
    for f in range(n):
        k = int(counts[f])
        xy = np.stack([un[f, :k]["x"], un[f, :k]["y"]], 1)
        on, om, oqm = oracle.search_local_map(xy, un[f, :k]["octave"], desc[f, :k], None, None, cs[f], ci[f, :k], ex.GetScaleFactors(),
                                              grid[0], grid[1], grid[2], grid[3], th, 0.8, want[f])
        assert nm[f] == on and on > 0.15 * k, (f, nm[f], on, k)
        assert (qm[f, :nq[f]] == oqm).all() and (match[f, :k] == om).all(), f
    ex.close()
    for a in list(d.values()) + [d_pts, d_nq, d_pose, d_ow, d_q, d_nv, d_match, d_qm, d_nm]:
        a.free()


def test_fuse_projection_then_fuse_search(oracle):
    n = 2
    ex, cap, d, grid = _extract(n, 92)
    rng = np.random.default_rng(5)
    counts = d["n"].to_host()
    un = d["un"].to_host().reshape(n, cap * 28).view(oa.KP_DTYPE).reshape(n, cap)
    desc, cs, ci = d["desc"].to_host(), d["cs"].to_host(), d["ci"].to_host()
    extra, th = 200, 3.0
    q_cap = cap + extra
    points = np.zeros((n, q_cap), oa.FRUSTUM_POINT_DTYPE)
    nq, poses, centres = np.zeros(n, np.int32), np.zeros((n, 12), np.float32), np.zeros((n, 3), np.float32)
    for f in range(n):
        T, Ow = _pose(rng, big=(f == 1))
        poses[f], centres[f] = T.ravel(), Ow
        k = int(counts[f])
        pts = np.concatenate([_points_from_keypoints(rng, T, Ow, un[f, :k], desc[f, :k], NL), _edge_points(rng, T, Ow, extra)])
        pts = pts[rng.permutation(len(pts))]
        nq[f] = len(pts)
        points[f, :len(pts)] = pts
    d_pts = DeviceArray.from_host(points.view(np.uint8).reshape(n, q_cap * 68))
    d_nq, d_pose, d_ow = DeviceArray.from_host(nq), DeviceArray.from_host(poses), DeviceArray.from_host(centres)
    d_q = DeviceArray((n, q_cap, 56), np.uint8, fill_byte=0x77)
    d_np = DeviceArray((n,), np.int32, fill_byte=0x55)
    prm = _params(0.5, th)
    ex.fuse_project_batch_device(n, prm, d_pose.ptr, d_ow.ptr, d_pts.ptr, q_cap, d_nq.ptr, d_q.ptr, d_np.ptr, sync=True)
    got = d_q.to_host().reshape(n, q_cap * 56).view(oa.WINDOW_QUERY_DTYPE).reshape(n, q_cap)
    npj = d_np.to_host()
    scale, inv_sigma2 = ex.GetScaleFactors(), ex.GetInverseScaleSigmaSquares()
    wants = []
    for f in range(n):
        u, v, ur, rad, lvl, ok = oracle.fuse_project(poses[f], centres[f], CAM, BOUNDS, prm.log_scale_factor, NL, scale, th,
                                                     points[f, :nq[f]])
        wants.append((u, v, ur, rad, lvl, ok))
        g = got[f, :nq[f]]
        assert npj[f] == ok.sum() and ok.sum() > 0.3 * counts[f], (f, npj[f], ok.sum())
        assert ((g["radius"] == -1.0) == ~ok).all()
        for name, w in (("u", u), ("v", v), ("ur", ur), ("radius", rad)):
            assert (g[name][ok].view(np.uint32) == w[ok].view(np.uint32)).all(), (f, name)
        assert (g["max_level"][ok] == lvl[ok]).all() and (g["min_level"][ok] == lvl[ok] - 1).all()
        assert (g["desc"] == points[f, :nq[f]]["desc"]).all()
    # the records are the queries of the Fuse search; a dropped point finds nothing
    d_bi, d_bd = DeviceArray((n, q_cap), np.int32, fill_byte=0x55), DeviceArray((n, q_cap), np.int32, fill_byte=0x55)
    ex.fuse_search_batch_device(n, d["un"].ptr, cap, d["cs"].ptr, d["ci"].ptr, d["desc"].ptr, 0, grid[0], grid[1], grid[2], grid[3],
                                d_q.ptr, q_cap, d_nq.ptr, d_bi.ptr, d_bd.ptr, sync=True)
    bi, bd = d_bi.to_host(), d_bd.to_host()
    found = 0
    for f in range(n):
        k = int(counts[f])
        xy = np.stack([un[f, :k]["x"], un[f, :k]["y"]], 1)
        u, v, ur, rad, lvl, ok = wants[f]
        for i in range(nq[f]):
            if not ok[i]:
                assert (bi[f, i], bd[f, i]) == (-1, 256), (f, i)
                continue
            idx, dist = oracle.fuse_search(xy, un[f, :k]["octave"], desc[f, :k], None, cs[f], ci[f, :k], grid[0], grid[1], grid[2],
                                           grid[3], float(u[i]), float(v[i]), float(ur[i]), float(rad[i]), int(lvl[i]), inv_sigma2,
                                           points[f, i]["desc"])
            assert (bi[f, i], bd[f, i]) == (idx, dist), (f, i)
            found += idx >= 0
    assert found > 0.2 * counts.sum()
    ex.close()
    for a in list(d.values()) + [d_pts, d_nq, d_pose, d_ow, d_q, d_np, d_bi, d_bd]:
        a.free()


@pytest.mark.parametrize("relocalisation,th,max_dist,check", [(True, 10.0, 100, True), (False, 4.0, 50, False)])
def test_guided_projection_then_guided_search(oracle, relocalisation, th, max_dist, check):
    n = 2
    ex, cap, d, grid = _extract(n, 93)
    rng = np.random.default_rng(7 + relocalisation)
    counts = d["n"].to_host()
    un = d["un"].to_host().reshape(n, cap * 28).view(oa.KP_DTYPE).reshape(n, cap)
    desc, cs, ci = d["desc"].to_host(), d["cs"].to_host(), d["ci"].to_host()
    extra = 200
    q_cap = cap + extra
    points = np.zeros((n, q_cap), oa.FRUSTUM_POINT_DTYPE)
    angles = np.zeros((n, q_cap), np.float32)
    nq, poses, centres = np.zeros(n, np.int32), np.zeros((n, 12), np.float32), np.zeros((n, 3), np.float32)
    for f in range(n):
        T, Ow = _pose(rng)
        poses[f], centres[f] = T.ravel(), Ow
        k = int(counts[f])
        pts = np.concatenate([_points_from_keypoints(rng, T, Ow, un[f, :k], desc[f, :k], NL), _edge_points(rng, T, Ow, extra)])
        ang = np.concatenate([(un[f, :k]["angle"] + rng.normal(0, 3, k)) % 360, rng.uniform(0, 360, extra)]).astype(np.float32)
        order = rng.permutation(len(pts))
        nq[f] = len(pts)
        points[f, :len(pts)], angles[f, :len(pts)] = pts[order], ang[order]
    d_pts = DeviceArray.from_host(points.view(np.uint8).reshape(n, q_cap * 68))
    d_ang = DeviceArray.from_host(angles)
    d_nq, d_pose, d_ow = DeviceArray.from_host(nq), DeviceArray.from_host(poses), DeviceArray.from_host(centres)
    d_q = DeviceArray((n, q_cap, 56), np.uint8, fill_byte=0x77)
    d_np = DeviceArray((n,), np.int32, fill_byte=0x55)
    prm = _params(0.5, th)
    ex.project_guided_batch_device(n, prm, d_pose.ptr, d_ow.ptr, d_pts.ptr, d_ang.ptr, q_cap, d_nq.ptr, relocalisation, d_q.ptr,
                                   d_np.ptr, sync=True)
    got = d_q.to_host().reshape(n, q_cap * 56).view(oa.GUIDED_QUERY_DTYPE).reshape(n, q_cap)
    npj = d_np.to_host()
    want = []
    for f in range(n):
        w, cnt = oracle.project_guided(poses[f], centres[f], CAM, BOUNDS, prm.log_scale_factor, NL, ex.GetScaleFactors(), th,
                                       points[f, :nq[f]], angles[f, :nq[f]], relocalisation)
        want.append(w)
        assert npj[f] == cnt and cnt > 0.3 * counts[f], (f, npj[f], cnt)
        assert _same_bits(got[f, :nq[f]], w, ("u", "v", "angle", "level", "radius", "flags", "desc")) is None, f
    # behind the camera: kept by the relocalisation's projection only
    Tz = poses.reshape(n, 3, 4)
    behind = sum(int(((Tz[f, 2, :3] * np.stack([points[f]["x"], points[f]["y"], points[f]["z"]], 1)).sum(1) + Tz[f, 2, 3] < 0)
                     [:nq[f]][want[f]["flags"] == 1].sum()) for f in range(n))
    assert (behind > 0) == bool(relocalisation)
    # the records are the queries of the guided search
    d_match, d_qm = DeviceArray((n, cap), np.int32, fill_byte=0x55), DeviceArray((n, q_cap), np.int32, fill_byte=0x55)
    d_nm = DeviceArray((n,), np.int32, fill_byte=0x55)
    level_up = 1 if relocalisation else 0
    ex.search_guided_batch_device(n, d["un"].ptr, cap, d["cs"].ptr, d["ci"].ptr, d["desc"].ptr, 0, grid[0], grid[1], grid[2], grid[3],
                                  d_q.ptr, q_cap, d_nq.ptr, level_up, max_dist, check, d_match.ptr, d_qm.ptr, d_nm.ptr, sync=True)
    match, qm, nm = d_match.to_host(), d_qm.to_host(), d_nm.to_host()
    for f in range(n):
        k = int(counts[f])
        xy = np.stack([un[f, :k]["x"], un[f, :k]["y"]], 1)
        on, om, oqm = oracle.search_guided(xy, un[f, :k]["octave"], un[f, :k]["angle"], desc[f, :k], None, cs[f], ci[f, :k], grid[0],
                                           grid[1], grid[2], grid[3], want[f], level_up, max_dist, check)
        assert nm[f] == on and on > 0.15 * k, (f, nm[f], on, k)
        assert (qm[f, :nq[f]] == oqm).all() and (match[f, :k] == om).all(), f
    ex.close()
    for a in list(d.values()) + [d_pts, d_ang, d_nq, d_pose, d_ow, d_q, d_np, d_match, d_qm, d_nm]:
        a.free()


def test_sim3_projection_then_window_search(oracle):
    n = 2
    ex, cap, d, grid = _extract(n, 94)
    rng = np.random.default_rng(11)
    counts = d["n"].to_host()
    un = d["un"].to_host().reshape(n, cap * 28).view(oa.KP_DTYPE).reshape(n, cap)
    desc, cs, ci = d["desc"].to_host(), d["cs"].to_host(), d["ci"].to_host()
    extra, th = 200, 7.5
    q_cap = cap + extra
    points = np.zeros((n, q_cap), oa.FRUSTUM_POINT_DTYPE)
    nq, poses = np.zeros(n, np.int32), np.zeros((n, 24), np.float32)
    for f in range(n):
        # camera 2 sees the points at pose T2 (where they project onto its keypoints); camera 1 and the similarity are
        # chosen so that sR21 * (R1w P + t1w) + t21 is about R2w P + t2w with a scale drift of s
        T2, Ow2 = _pose(rng)
        T1, _ = _pose(rng, big=True)
        s21 = 1.0 + 0.05 * f
        R1, t1 = T1[:, :3].astype(np.float64), T1[:, 3].astype(np.float64)
        R2, t2 = T2[:, :3].astype(np.float64), T2[:, 3].astype(np.float64)
        R21 = R2 @ R1.T
        S = np.concatenate([s21 * R21, (s21 * (t2 - R21 @ t1))[:, None]], 1).astype(np.float32)
        poses[f, :12], poses[f, 12:] = T1.ravel(), S.ravel()
        k = int(counts[f])
        pts = _points_from_keypoints(rng, T2, Ow2, un[f, :k], desc[f, :k], NL)
        pts["max_dist"] *= s21
        pts["min_dist"] *= s21
        pts = np.concatenate([pts, _edge_points(rng, T2, Ow2, extra)])
        pts = pts[rng.permutation(len(pts))]
        nq[f] = len(pts)
        points[f, :len(pts)] = pts
    d_pts = DeviceArray.from_host(points.view(np.uint8).reshape(n, q_cap * 68))
    d_nq, d_pose = DeviceArray.from_host(nq), DeviceArray.from_host(poses)
    d_q = DeviceArray((n, q_cap, 56), np.uint8, fill_byte=0x77)
    d_np = DeviceArray((n,), np.int32, fill_byte=0x55)
    prm = _params(0.5, th)
    ex.sim3_project_batch_device(n, prm, d_pose.ptr, d_pts.ptr, q_cap, d_nq.ptr, d_q.ptr, d_np.ptr, sync=True)
    got = d_q.to_host().reshape(n, q_cap * 56).view(oa.WINDOW_QUERY_DTYPE).reshape(n, q_cap)
    npj = d_np.to_host()
    wants = []
    for f in range(n):
        u, v, rad, lvl, ok = oracle.sim3_project(poses[f, :12], poses[f, 12:], CAM, BOUNDS, prm.log_scale_factor, NL,
                                                 ex.GetScaleFactors(), th, points[f, :nq[f]])
        wants.append((u, v, rad, lvl, ok))
        g = got[f, :nq[f]]
        assert npj[f] == ok.sum() and ok.sum() > 0.5 * counts[f], (f, npj[f], ok.sum())
        assert ((g["radius"] == -1.0) == ~ok).all()
        for name, w in (("u", u), ("v", v), ("radius", rad)):
            assert (g[name][ok].view(np.uint32) == w[ok].view(np.uint32)).all(), (f, name)
        assert (g["max_level"][ok] == lvl[ok]).all() and (g["min_level"][ok] == lvl[ok] - 1).all()
        assert (g["desc"] == points[f, :nq[f]]["desc"]).all()
    d_bi, d_bd = DeviceArray((n, q_cap), np.int32, fill_byte=0x55), DeviceArray((n, q_cap), np.int32, fill_byte=0x55)
    ex.search_window_batch_device(n, d["un"].ptr, cap, d["cs"].ptr, d["ci"].ptr, d["desc"].ptr, 0, 0, grid[0], grid[1], grid[2],
                                  grid[3], d_q.ptr, q_cap, d_nq.ptr, d_bi.ptr, d_bd.ptr, sync=True)
    bi, bd = d_bi.to_host(), d_bd.to_host()
    found = 0
    for f in range(n):
        k = int(counts[f])
        xy = np.stack([un[f, :k]["x"], un[f, :k]["y"]], 1)
        u, v, rad, lvl, ok = wants[f]
        assert ((bi[f, :nq[f]][~ok] == -1) & (bd[f, :nq[f]][~ok] == 256)).all()
        for i in np.flatnonzero(ok)[::2]:
            idx, dist = oracle.search_window(xy, un[f, :k]["octave"], desc[f, :k], None, None, cs[f], ci[f, :k], grid[0], grid[1],
                                             grid[2], grid[3], float(u[i]), float(v[i]), float(rad[i]), int(lvl[i]) - 1, int(lvl[i]),
                                             0.0, points[f, i]["desc"])[:2]
            assert (bi[f, i], bd[f, i]) == (idx, dist), (f, i)
            found += idx >= 0 and dist <= 100
    assert found > 0.15 * counts.sum()
    ex.close()
    for a in list(d.values()) + [d_pts, d_nq, d_pose, d_q, d_np, d_bi, d_bd]:
        a.free()


def test_projection_rejects_bad_arguments():
    ex = oa.ORBextractor(500, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=1, keep_pyramid=False)
    with pytest.raises(oa.OrbxError):
        ex.is_in_frustum_batch_device(1, _params(), 0, 0, 0, 16, 0, 0, 0)
    with pytest.raises(oa.OrbxError):
        ex.fuse_project_batch_device(0, _params(), 1, 1, 1, 16, 1, 1, 1)
    ex.close()
