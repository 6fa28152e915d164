"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them bit for bit."""
import glob
import hashlib
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ALL = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
FILES = [f for f in ALL if not os.path.basename(f).startswith(("projection_", "rgbd_"))]   # extractor fixtures
PROJ = [f for f in ALL if os.path.basename(f).startswith("projection_")]           # SearchByProjection fixtures
RGBD = [f for f in ALL if os.path.basename(f).startswith("rgbd_")]                 # ComputeStereoFromRGBD fixtures


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _same(kps, desc, g):
    assert len(kps) == len(g["keypoints"])
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert (kps[f] == g["keypoints"][f]).all(), f
    assert (desc == g["descriptors"]).all()


def test_fixtures_present():
    assert len(FILES) >= 2


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path, allow_pickle=False)
    nl = int(g["nlevels"])
    orc = oracle.Oracle(int(g["nfeatures"]), float(g["scale_factor"]), nl, int(g["ini_th"]), int(g["min_th"]))
    kps, desc = orc.extract(g["image"])
    _same(kps, desc, g)
    assert [_sha(orc.level(l)) for l in range(nl)] == list(g["level_sha256"])
    assert [_sha(orc.blurred(l)) for l in range(nl)] == list(g["blurred_sha256"])
    assert [len(orc.candidates(l)) for l in range(nl)] == list(g["candidates_per_level"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_reproduces_golden(path):
    import orb_slam2_amd as oa
    g = np.load(path, allow_pickle=False)
    nl = int(g["nlevels"])
    img = g["image"]
    ex = oa.ORBextractor(int(g["nfeatures"]), float(g["scale_factor"]), nl, int(g["ini_th"]), int(g["min_th"]),
                         max_width=img.shape[1], max_height=img.shape[0])
    kps, desc = ex(img)
    _same(kps, desc, g)
    for l in range(nl):
        assert _sha(ex.pyramid_level(0, l)) == g["level_sha256"][l]
        w, h = ex.pyramid_level(0, l).shape[::-1]
        assert _sha(ex.debug_read(oa.DBG_BLURRED, 0, l).reshape(h, w)) == g["blurred_sha256"][l]
        assert len(ex.debug_read(oa.DBG_CANDIDATES, 0, l)) // 12 == g["candidates_per_level"][l]
        assert len(ex.debug_read(oa.DBG_LEVEL_KPS, 0, l)) // 12 == g["keypoints_per_level"][l]
    ex.close()


# ---- SearchByProjection between two consecutive real RGB-D frames (tests/golden/projection_*.npz) ----
def _projection_inputs(g):
    import oracle_lib
    P = np.frombuffer(g["params"].tobytes(), oracle_lib.PROJ_PARAMS_DTYPE)[0]
    q = np.frombuffer(g["queries"].tobytes(), oracle_lib.MAP_QUERY_DTYPE)
    eye = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32)
    return P, q, eye


def test_projection_fixture_present():
    assert len(PROJ) >= 1


@pytest.mark.parametrize("path", PROJ, ids=[os.path.basename(f) for f in PROJ])
def test_oracle_reproduces_projection_golden(oracle, path):
    g = np.load(path, allow_pickle=False)
    P, q, eye = _projection_inputs(g)
    orc = oracle.Oracle(1000, 1.2, 8, 20, 7)
    kps, desc = orc.extract(g["image"])
    un = oracle.undistort_points(np.stack([kps["x"], kps["y"]], 1).astype(np.float32), g["K"], g["dist"])
    start, idx = oracle.assign_grid(un, float(P["min_x"]), float(P["min_y"]), float(P["inv_w"]), float(P["inv_h"]))
    n, match, qm = oracle.search_by_projection(un, kps["octave"].astype(np.int32), kps["angle"], desc, g["u_right"], None, start,
                                               idx, np.array(orc.params.scale[:8], np.float32), P, eye, eye, q)
    assert n == int(g["nmatches"]) and n > 400
    assert np.array_equal(match, g["match"]) and np.array_equal(qm, g["query_match"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", PROJ, ids=[os.path.basename(f) for f in PROJ])
def test_hip_reproduces_projection_golden(path):
    """The device chain extract -> undistort + grid -> SearchByProjection on the fixture's current frame."""
    import orb_slam2_amd as oa
    from hip_rt import DeviceArray
    g = np.load(path, allow_pickle=False)
    P, q, eye = _projection_inputs(g)
    img = g["image"]
    h, w = img.shape
    ex = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    q_cap = len(q)
    d_img = DeviceArray.from_host(img.reshape(1, h, w))
    d_kps, d_desc = DeviceArray((1, cap, 28), np.uint8, fill_byte=0), DeviceArray((1, cap, 32), np.uint8)
    d_n = DeviceArray((1,), np.int32, fill_byte=0)
    d_un = DeviceArray((1, cap, 28), np.uint8, fill_byte=0)
    d_cs, d_ci = DeviceArray((1, 64 * 48 + 1), np.int32, fill_byte=0), DeviceArray((1, cap), np.int32, fill_byte=0)
    ur = np.full((1, cap), -1, np.float32)
    ur[0, :len(g["u_right"])] = g["u_right"]
    d_ur = DeviceArray.from_host(ur)
    d_q = DeviceArray.from_host(np.ascontiguousarray(g["queries"]).reshape(1, q_cap * 56))
    d_nq = DeviceArray.from_host(np.array([q_cap], np.int32))
    d_poses = DeviceArray.from_host(np.concatenate([eye, eye]).reshape(1, 24))
    d_match, d_qm = DeviceArray((1, cap), np.int32, fill_byte=0x55), DeviceArray((1, q_cap), np.int32, fill_byte=0x55)
    d_nm = DeviceArray((1,), np.int32, fill_byte=0x55)
    ex.extract_batch_device(d_img.ptr, 1, w, h, w, h * w, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=False)
    ex.undistort_grid_batch_device(1, d_kps.ptr, cap, d_n.ptr, g["K"], g["dist"], float(P["min_x"]), float(P["min_y"]),
                                   float(P["inv_w"]), float(P["inv_h"]), d_un.ptr, d_cs.ptr, d_ci.ptr, sync=False)
    Pd = np.zeros((), oa.PROJECTION_PARAMS_DTYPE)
    for k in Pd.dtype.names:
        Pd[k] = P[k]
    ex.search_by_projection_batch_device(1, d_un.ptr, cap, d_cs.ptr, d_ci.ptr, d_desc.ptr, d_ur.ptr, 0, Pd, d_poses.ptr, d_q.ptr,
                                         q_cap, d_nq.ptr, d_match.ptr, d_qm.ptr, d_nm.ptr, sync=True)
    k = int(d_n.to_host()[0])
    assert k == len(g["match"])
    assert int(d_nm.to_host()[0]) == int(g["nmatches"])
    assert np.array_equal(d_match.to_host()[0, :k], g["match"])
    assert np.array_equal(d_qm.to_host()[0], g["query_match"])
    ex.close()
    for a in (d_img, d_kps, d_desc, d_n, d_un, d_cs, d_ci, d_ur, d_q, d_nq, d_poses, d_match, d_qm, d_nm):
        a.free()


# ---- Frame::ComputeStereoFromRGBD on a real RGB-D frame (tests/golden/rgbd_*.npz: the raw 16-bit depth map of
# ---- the reference's bundled TUM sequence) ----
def test_rgbd_fixture_present():
    assert len(RGBD) >= 1


@pytest.mark.parametrize("path", RGBD, ids=[os.path.basename(f) for f in RGBD])
def test_oracle_reproduces_rgbd_golden(oracle, path):
    g = np.load(path, allow_pickle=False)
    orc = oracle.Oracle(1000, 1.2, 8, 20, 7)
    kps, _ = orc.extract(g["image"])
    assert len(kps) == int(g["n"])
    xy = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
    un = oracle.undistort_points(xy, g["K"], g["dist"])
    ur, z = oracle.stereo_from_rgbd(g["depth"], float(g["factor"]), xy, un[:, 0], float(g["mbf"]))
    assert (ur.view(np.uint32) == g["u_right"].view(np.uint32)).all() and (z.view(np.uint32) == g["depth_out"].view(np.uint32)).all()
    # the depth the fixture stores is the sample times the float reciprocal of 5000, not the sample divided by 5000
    has = z > 0
    raw = g["depth"][kps["y"].astype(np.int32), kps["x"].astype(np.int32)].astype(np.float32)
    assert (z[has] == raw[has] * np.float32(g["factor"])).all() and has.sum() > 500


@pytest.mark.gpu
@pytest.mark.parametrize("path", RGBD, ids=[os.path.basename(f) for f in RGBD])
def test_hip_reproduces_rgbd_golden(path):
    import orb_slam2_amd as oa
    from hip_rt import DeviceArray
    g = np.load(path, allow_pickle=False)
    img, depth = g["image"], g["depth"]
    h, w = img.shape
    ex = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    d_img, d_depth = DeviceArray.from_host(img), DeviceArray.from_host(depth)
    d_kps, d_un = DeviceArray((1, cap), oa.KP_DTYPE), DeviceArray((1, cap), oa.KP_DTYPE)
    d_desc = DeviceArray((1, cap, 32), np.uint8)
    d_n = DeviceArray((1,), np.int32, fill_byte=0)
    d_cs, d_ci = DeviceArray((1, 64 * 48 + 1), np.int32), DeviceArray((1, cap), np.int32)
    d_ur, d_z = DeviceArray((1, cap), np.float32, fill_byte=0x55), DeviceArray((1, cap), np.float32, fill_byte=0x55)
    ex.extract_batch_device(d_img.ptr, 1, w, h, w, w * h, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=False)
    ex.undistort_grid_batch_device(1, d_kps.ptr, cap, d_n.ptr, g["K"], g["dist"], 0.0, 0.0, 0.1, 0.1, d_un.ptr, d_cs.ptr, d_ci.ptr,
                                   sync=False)
    ex.rgbd_depth_batch_device(1, d_depth.ptr, False, w, h, 2 * w, 2 * w * h, float(g["factor"]), d_kps.ptr, d_un.ptr, cap,
                               d_n.ptr, float(g["mbf"]), d_ur.ptr, d_z.ptr, sync=True)
    n = int(d_n.to_host()[0])
    assert n == int(g["n"])
    ur, z = d_ur.to_host()[0], d_z.to_host()[0]
    assert (ur[:n].view(np.uint32) == g["u_right"].view(np.uint32)).all()
    assert (z[:n].view(np.uint32) == g["depth_out"].view(np.uint32)).all()
    assert (ur[n:] == -1).all() and (z[n:] == -1).all()
    ex.close()
    for a in (d_img, d_depth, d_kps, d_un, d_desc, d_n, d_cs, d_ci, d_ur, d_z):
        a.free()
