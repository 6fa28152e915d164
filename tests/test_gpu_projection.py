"""GPU parity of the whole ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) (reference
src/ORBmatcher.cc:1365-1507) against the CPU oracle's sequential loop, bit-exact: the feature each map
point is written to, the map point each feature ends with after the rotation check, and nmatches."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth
from hip_rt import DeviceArray

pytestmark = pytest.mark.gpu

K = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)            # TUM1.yaml
DIST = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)


def small_pose(rng, t_scale, rot_deg):
    """Rows of [R|t]: a rotation of rot_deg about a random axis, a translation of length ~t_scale."""
    a = rng.normal(size=3)
    a /= np.linalg.norm(a)
    th = np.deg2rad(rot_deg)
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = rng.normal(size=3) * t_scale
    return np.concatenate([R, t[:, None]], 1).astype(np.float32).reshape(12)


@pytest.mark.parametrize("mono,check,use_right,use_taken,th,dz", [
    (0, 1, True, True, 15.0, 0.0),     # RGB-D / stereo tracking as TrackWithMotionModel calls it
    (1, 1, False, False, 15.0, 0.0),   # monocular: th 15, no right coordinates
    (0, 0, True, False, 30.0, 0.6),    # forward motion beyond the baseline: levels >= the last octave, wide windows
    (0, 1, False, True, 7.0, -0.6),    # backward motion: levels <= the last octave
])
def test_search_by_projection_after_extraction(oracle, mono, check, use_right, use_taken, th, dz):
    w, h, nf, nl, n = 640, 480, 1000, 8, 3
    imgs = synth.make_batch(n, w, h, 1, 120)
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
    rng = np.random.default_rng(23 + mono + 2 * check)
    P = np.zeros((), oa.PROJECTION_PARAMS_DTYPE)
    P["fx"], P["fy"], P["cx"], P["cy"] = K
    P["mbf"], P["mb"] = 40.0, 40.0 / K[0]
    P["min_x"], P["min_y"], P["max_x"], P["max_y"], P["inv_w"], P["inv_h"] = min_x, min_y, max_x, max_y, inv_w, inv_h
    P["th"], P["mono"], P["check_orientation"] = th, mono, check
    # The "last frame" of a pair is the current frame itself seen from a slightly different pose: its map
    # points are the current features back-projected at random depths, so most of them project within a
    # few pixels of a feature with a near-identical descriptor, and neighbours compete for the same
    # features.  Some points appear twice (the same feature wanted by two points), some have no
    # observations (they do not hold what they take), some are skipped.
    q_cap = cap + 64
    queries = np.zeros((n, q_cap), oa.MAP_QUERY_DTYPE)
    nq = np.zeros(n, np.int32)
    poses = np.zeros((n, 24), np.float32)
    for f in range(n):
        k = int(counts[f])
        Tlw = small_pose(rng, 0.0, 0.0)
        Tcw = small_pose(rng, 0.01, 0.15)
        Tlw[11] += dz        # tlc.z = dz when the current camera sits at the origin: forward / backward
        poses[f, :12], poses[f, 12:] = Tcw, Tlw
        order = rng.permutation(k)
        pick = np.concatenate([order, rng.choice(k, 60)])[:q_cap]
        src = un[f, pick]
        z = rng.uniform(1.5, 12.0, len(pick))
        q = queries[f]
        m = len(pick)
        q["x"][:m] = (src["x"] - K[2]) * z / K[0]
        q["y"][:m] = (src["y"] - K[3]) * z / K[1]
        q["z"][:m] = z
        q["z"][:m][rng.random(m) < 0.02] *= -1            # behind the camera
        q["octave"][:m] = src["octave"]
        q["angle"][:m] = np.mod(src["angle"] + rng.normal(0, 25, m), 360).astype(np.float32)
        fl = rng.random(m)
        q["flags"][:m] = np.where(fl < 0.08, 2, np.where(fl < 0.30, 0, 1))
        q["desc"][:m] = desc[f, pick]
        flip = rng.random((m, 32)) < 0.04
        q["desc"][:m] ^= (flip * rng.integers(1, 256, (m, 32))).astype(np.uint8)
        nq[f] = m
    u_right = np.where(rng.random((n, cap)) < 0.6, un["x"] - rng.uniform(2, 30, (n, cap)), -1).astype(np.float32)
    taken = (rng.random((n, cap)) < 0.1).astype(np.uint8)
    d_q = DeviceArray.from_host(queries.view(np.uint8).reshape(n, q_cap * 56))
    d_nq, d_poses = DeviceArray.from_host(nq), DeviceArray.from_host(poses)
    d_ur, d_tk = DeviceArray.from_host(u_right), DeviceArray.from_host(taken)
    d_match = DeviceArray((n, cap), np.int32, fill_byte=0x55)
    d_qm = DeviceArray((n, q_cap), np.int32, fill_byte=0x55)
    d_nm = DeviceArray((n,), np.int32, fill_byte=0x55)
    ex.search_by_projection_batch_device(n, d_un.ptr, cap, d_cs.ptr, d_ci.ptr, d_desc.ptr, d_ur.ptr if use_right else 0,
                                         d_tk.ptr if use_taken else 0, P, d_poses.ptr, d_q.ptr, q_cap, d_nq.ptr,
                                         d_match.ptr, d_qm.ptr, d_nm.ptr, sync=True)
    match, qm, nm = d_match.to_host(), d_qm.to_host(), d_nm.to_host()
    sf = np.array(ex.GetScaleFactors(), np.float32)
    total = 0
    for f in range(n):
        k = int(counts[f])
        xy = np.stack([un[f, :k]["x"], un[f, :k]["y"]], 1)
        e_n, e_match, e_qm = oracle.search_by_projection(
            xy, un[f, :k]["octave"].astype(np.int32), un[f, :k]["angle"], desc[f, :k], u_right[f, :k] if use_right else None,
            taken[f, :k] if use_taken else None, cs[f], ci[f], sf, P, poses[f, :12], poses[f, 12:], queries[f, :nq[f]])
        assert np.array_equal(qm[f, :nq[f]], e_qm), "frame %d: first difference at point %d" % (
            f, int(np.flatnonzero(qm[f, :nq[f]] != e_qm)[0]))
        assert (qm[f, nq[f]:] == -1).all()
        assert np.array_equal(match[f, :k], e_match)
        assert (match[f, k:] == -1).all()
        assert nm[f] == e_n
        total += e_n
        # the sequential dependence is exercised: some point did not get the feature it would get alone
        alone = [oracle.search_by_projection(xy, un[f, :k]["octave"].astype(np.int32), un[f, :k]["angle"], desc[f, :k],
                                             u_right[f, :k] if use_right else None, taken[f, :k] if use_taken else None,
                                             cs[f], ci[f], sf, P, poses[f, :12], poses[f, 12:], queries[f, j:j + 1])[2][0]
                 for j in range(0, nq[f], 7)]
        assert dz != 0.0 or th < 10 or any(a != b for a, b in zip(alone, e_qm[::7]))
    assert total > 0.3 * nq.sum()
    ex.close()
    for a in (d_img, d_kps, d_desc, d_n, d_un, d_cs, d_ci, d_q, d_nq, d_poses, d_ur, d_tk, d_match, d_qm, d_nm):
        a.free()


def _run_handmade(oracle, frames, P, q_cap=None):
    """frames: list of dicts(xy, octave, angle, desc, queries, Tcw, Tlw[, u_right, taken]); one batch through the
    C ABI, each frame compared with the oracle."""
    n = len(frames)
    cap = max(len(F["xy"]) for F in frames) + 3
    q_cap = q_cap or max(len(F["queries"]) for F in frames) + 5
    kps = np.zeros((n, cap), oa.KP_DTYPE)
    desc = np.zeros((n, cap, 32), np.uint8)
    cs = np.zeros((n, 64 * 48 + 1), np.int32)
    ci = np.full((n, cap), -1, np.int32)
    queries = np.zeros((n, q_cap), oa.MAP_QUERY_DTYPE)
    nq = np.zeros(n, np.int32)
    poses = np.zeros((n, 24), np.float32)
    u_right = np.full((n, cap), -1, np.float32)
    taken = np.zeros((n, cap), np.uint8)
    for f, F in enumerate(frames):
        k = len(F["xy"])
        kps[f, :k]["x"], kps[f, :k]["y"] = F["xy"][:, 0], F["xy"][:, 1]
        kps[f, :k]["octave"], kps[f, :k]["angle"] = F["octave"], F["angle"]
        desc[f, :k] = F["desc"]
        s, i = oracle.assign_grid(F["xy"], float(P["min_x"]), float(P["min_y"]), float(P["inv_w"]), float(P["inv_h"]))
        cs[f], ci[f, :len(i)] = s, i
        queries[f, :len(F["queries"])] = F["queries"]
        nq[f] = len(F["queries"])
        poses[f, :12], poses[f, 12:] = F["Tcw"], F["Tlw"]
        if "u_right" in F:
            u_right[f, :k] = F["u_right"]
        if "taken" in F:
            taken[f, :k] = F["taken"]
    ex = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1, keep_pyramid=False)
    dev = [DeviceArray.from_host(a) for a in (kps.view(np.uint8).reshape(n, cap * 28), cs, ci, desc, u_right, taken, poses,
                                              queries.view(np.uint8).reshape(n, q_cap * 56), nq)]
    d_match = DeviceArray((n, cap), np.int32, fill_byte=0x55)
    d_qm = DeviceArray((n, q_cap), np.int32, fill_byte=0x55)
    d_nm = DeviceArray((n,), np.int32, fill_byte=0x55)
    ex.search_by_projection_batch_device(n, dev[0].ptr, cap, dev[1].ptr, dev[2].ptr, dev[3].ptr, dev[4].ptr, dev[5].ptr, P,
                                         dev[6].ptr, dev[7].ptr, q_cap, dev[8].ptr, d_match.ptr, d_qm.ptr, d_nm.ptr, sync=True)
    match, qm, nm = d_match.to_host(), d_qm.to_host(), d_nm.to_host()
    sf = np.array(ex.GetScaleFactors(), np.float32)
    out = []
    for f, F in enumerate(frames):
        k = len(F["xy"])
        e = oracle.search_by_projection(F["xy"], F["octave"], F["angle"], F["desc"], u_right[f, :k], taken[f, :k], cs[f], ci[f],
                                        sf, P, F["Tcw"], F["Tlw"], F["queries"])
        assert np.array_equal(qm[f, :nq[f]], e[2]), "frame %d" % f
        assert np.array_equal(match[f, :k], e[1]), "frame %d" % f
        assert nm[f] == e[0]
        assert (match[f, k:] == -1).all() and (qm[f, nq[f]:] == -1).all()
        out.append(e)
    ex.close()
    for a in dev + [d_match, d_qm, d_nm]:
        a.free()
    return out


def _params(th, check, mono=0):
    P = np.zeros((), oa.PROJECTION_PARAMS_DTYPE)
    P["fx"], P["fy"], P["cx"], P["cy"], P["mbf"], P["mb"] = 100.0, 100.0, 320.0, 240.0, 10.0, 0.1
    P["min_x"], P["min_y"], P["max_x"], P["max_y"], P["inv_w"], P["inv_h"] = 0.0, 0.0, 640.0, 480.0, 0.1, 0.1
    P["th"], P["mono"], P["check_orientation"] = th, mono, check
    return P


EYE = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32)


@pytest.mark.parametrize("check", [0, 1])
def test_projection_long_dependency_chains(oracle, check):
    """Crowds: hundreds of map points aimed at the same spot, every feature acceptable to every point, so each
    point's choice depends on what all earlier holders took -- the worst case for the concurrent rounds."""
    rng = np.random.default_rng(5 + check)
    frames = []
    for k, m, hold in [(300, 340, 1.0), (150, 400, 0.7), (40, 25, 0.5), (1, 3, 1.0), (64, 64, 0.0)]:
        xy = (np.array([200.0, 150.0]) + rng.uniform(-6, 6, (k, 2))).astype(np.float32)
        desc = (rng.random((k, 32)) < 0.12).astype(np.uint8)                     # few bits set: every distance <= 100
        q = np.zeros(m, oa.MAP_QUERY_DTYPE)
        z = rng.uniform(2, 5, m).astype(np.float32)
        q["x"], q["y"], q["z"] = (200.0 - 320.0 + rng.uniform(-1, 1, m)) * z / 100.0, (150.0 - 240.0) * z / 100.0, z
        q["octave"] = rng.integers(0, 3, m)
        q["angle"] = rng.uniform(0, 360, m)
        q["flags"] = (rng.random(m) < hold).astype(np.int32)
        q["desc"] = (rng.random((m, 32)) < 0.12).astype(np.uint8)
        frames.append(dict(xy=xy, octave=rng.integers(0, 4, k).astype(np.int32), angle=rng.uniform(0, 360, k).astype(np.float32),
                           desc=desc, queries=q, Tcw=EYE, Tlw=EYE))
    # check = 1 also asks for a point capacity whose round state no longer fits the workgroup's LDS: the
    # kernel then keeps it in global scratch memory
    out = _run_handmade(oracle, frames, _params(12.0, check), q_cap=9000 if check else None)
    assert out[0][0] > 100 or check           # the crowd really matched


def test_projection_known_answers_on_device(oracle):
    """The hand-derived cases of the oracle's known-answer tests, through the device path."""
    def feat(pts, octaves=None, angles=None, desc0=None):
        k = len(pts)
        d = np.zeros((k, 32), np.uint8)
        if desc0 is not None:
            d[:, 0] = desc0
        return dict(xy=np.array(pts, np.float32), octave=np.zeros(k, np.int32) if octaves is None else np.array(octaves, np.int32),
                    angle=np.zeros(k, np.float32) if angles is None else np.array(angles, np.float32), desc=d, Tcw=EYE, Tlw=EYE)

    def at(u, v, z=2.0, octave=0, angle=0.0, flags=1):
        q = np.zeros((), oa.MAP_QUERY_DTYPE)
        q["x"], q["y"], q["z"], q["octave"], q["angle"], q["flags"] = (u - 320.0) * z / 100.0, (v - 240.0) * z / 100.0, z, octave, angle, flags
        return q

    def Q(*qs):
        return np.array(list(qs), oa.MAP_QUERY_DTYPE)

    two = [(100, 100), (104, 100)]
    fwd, back = EYE.copy(), EYE.copy()
    fwd[11], back[11] = 0.5, -0.5
    lv = dict(feat([(100, 100)] * 5, octaves=[0, 1, 2, 3, 4], desc0=[0x1F, 0x07, 0x0F, 0x03, 0x01]))
    frames = [
        dict(feat(two, desc0=[0, 1]), queries=Q(at(102, 100), at(102, 100))),
        dict(feat(two, desc0=[0, 1]), queries=Q(at(102, 100, flags=0), at(102, 100))),
        dict(feat(two, desc0=[0, 1]), queries=Q(at(102, 100, flags=2), at(102, 100)), taken=np.array([1, 0], np.uint8)),
        dict(feat([(100, 100)]), queries=Q(at(100, 100, z=-2.0), at(700, 100))),
        dict(lv, queries=Q(at(100, 100, octave=2))),
        dict(lv, queries=Q(at(100, 100, octave=2)), Tlw=fwd),
        dict(lv, queries=Q(at(100, 100, octave=2)), Tlw=back),
        dict(feat([(100, 100), (103, 100)], desc0=[0, 1]), queries=Q(at(101, 100)), u_right=np.array([50.0, -1.0], np.float32)),
        dict(feat([(100, 100), (103, 100)], desc0=[0, 1]), queries=Q(at(101, 100)), u_right=np.array([106.0, -1.0], np.float32)),
    ]
    frames.append(dict(feat(two), queries=np.zeros(0, oa.MAP_QUERY_DTYPE)))                    # a last frame without map points
    frames.append(dict(feat([(100, 100)]), queries=Q(at(100, 100, flags=2), at(100, 100, flags=3))))  # only skipped entries
    out = _run_handmade(oracle, frames, _params(10.0, 0))
    assert [list(e[2]) for e in out] == [[0, 1], [0, 0], [-1, 1], [-1, -1], [3], [4], [1], [1], [0], [], [-1, -1]]
    assert out[9][0] == 0 and out[10][0] == 0
    assert list(out[1][1]) == [1, -1] and [e[0] for e in out[:4]] == [2, 2, 1, 0]
    pts = [(40 + 40 * i, 100) for i in range(14)]
    rot = dict(feat(pts, angles=[0.0] * 12 + [260.0, 10.0]), queries=Q(*[at(40 + 40 * i, 100) for i in range(14)]))
    out = _run_handmade(oracle, [rot], _params(10.0, 1))
    assert out[0][0] == 12 and list(out[0][1]) == list(range(12)) + [-1, -1]


def test_projection_rejects_bad_arguments():
    ex = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1, keep_pyramid=False)
    buf = DeviceArray((4096,), np.int32, fill_byte=0)
    P = _params(10.0, 1)
    good = dict(n=1, d_kps_un=buf.ptr, kp_cap=8, d_cell_start=buf.ptr, d_cell_idx=buf.ptr, d_desc=buf.ptr, d_u_right=0,
                d_taken=0, params=P, d_poses=buf.ptr, d_queries=buf.ptr, q_cap=4, d_nq=buf.ptr, d_match=buf.ptr,
                d_query_match=buf.ptr, d_nmatches=buf.ptr)
    for bad in (dict(n=0), dict(kp_cap=0), dict(q_cap=0), dict(d_queries=0), dict(d_match=0), dict(d_poses=0),
                dict(q_cap=1 << 30)):
        with pytest.raises(oa.OrbxError):
            ex.search_by_projection_batch_device(**dict(good, **bad))
    ex.close()
    buf.free()
