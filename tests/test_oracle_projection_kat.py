"""Known-answer tests of the oracle's ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)
(reference src/ORBmatcher.cc:1365-1507): hand-built frames whose result can be read off the source."""
import numpy as np

import oracle_lib

GRID = dict(min_x=0.0, min_y=0.0, inv_w=64 / 640.0, inv_h=48 / 480.0)
EYE = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32)
SCALE = np.array([1.2 ** i for i in range(8)], np.float32)


def params(th=10.0, mono=0, check=0, mb=0.1):
    P = np.zeros((), oracle_lib.PROJ_PARAMS_DTYPE)
    P["fx"], P["fy"], P["cx"], P["cy"] = 100.0, 100.0, 320.0, 240.0
    P["mbf"], P["mb"] = 10.0, mb
    P["min_x"], P["min_y"], P["max_x"], P["max_y"] = 0.0, 0.0, 640.0, 480.0
    P["inv_w"], P["inv_h"], P["th"] = GRID["inv_w"], GRID["inv_h"], th
    P["mono"], P["check_orientation"] = mono, check
    return P


def frame(oracle, pts, octaves=None, angles=None):
    xy = np.array(pts, np.float32)
    start, idx = oracle.assign_grid(xy, 0.0, 0.0, GRID["inv_w"], GRID["inv_h"])
    n = len(pts)
    return dict(xy=xy, octave=np.zeros(n, np.int32) if octaves is None else np.array(octaves, np.int32),
                angle=np.zeros(n, np.float32) if angles is None else np.array(angles, np.float32),
                desc=np.zeros((n, 32), np.uint8), start=start, idx=idx)


def at_pixel(u, v, z=2.0, octave=0, angle=0.0, flags=1, desc0=0):
    """A map point that the identity pose projects to (u, v): x = (u - cx) z / fx, exact for these values."""
    q = np.zeros((), oracle_lib.MAP_QUERY_DTYPE)
    q["x"], q["y"], q["z"] = (u - 320.0) * z / 100.0, (v - 240.0) * z / 100.0, z
    q["octave"], q["angle"], q["flags"] = octave, angle, flags
    q["desc"][0] = desc0
    return q


def run(F, P, queries, Tcw=EYE, Tlw=EYE, u_right=None, taken0=None):
    q = np.array(queries, oracle_lib.MAP_QUERY_DTYPE)
    return oracle_lib.search_by_projection(F["xy"], F["octave"], F["angle"], F["desc"], u_right, taken0, F["start"], F["idx"],
                                           SCALE, P, Tcw, Tlw, q)


def test_later_point_skips_a_held_feature(oracle):
    F = frame(oracle, [(100, 100), (104, 100)])
    F["desc"][1, 0] = 0x01
    n, match, qm = run(F, params(), [at_pixel(102, 100), at_pixel(102, 100)])
    assert n == 2 and list(match) == [0, 1] and list(qm) == [0, 1]      # the second point takes the next best
    # a first point without observations does not hold the feature: the second overwrites it, both count
    n, match, qm = run(F, params(), [at_pixel(102, 100, flags=0), at_pixel(102, 100)])
    assert n == 2 and list(match) == [1, -1] and list(qm) == [0, 0]
    # features held before the call are skipped; skipped entries of the last frame do nothing
    n, match, qm = run(F, params(), [at_pixel(102, 100, flags=2), at_pixel(102, 100)], taken0=np.array([1, 0], np.uint8))
    assert n == 1 and list(match) == [-1, 1] and list(qm) == [-1, 1]


def test_projection_gates(oracle):
    F = frame(oracle, [(100, 100)])
    assert run(F, params(), [at_pixel(100, 100, z=-2.0)])[0] == 0          # behind the camera (:1404)
    assert run(F, params(), [at_pixel(700, 100)])[0] == 0                   # outside the image bounds (:1410)
    q = at_pixel(100, 100)
    q["desc"][:13] = 0xFF                                                    # distance 104 > TH_HIGH
    assert run(F, params(), [q])[0] == 0
    q["desc"][12] = 0x0F                                                     # distance 100 passes (<=)
    assert run(F, params(), [q])[0] == 1
    # window half-size = th * scale factor of the last frame's octave (:1418): 12 px away, th 10
    far = at_pixel(112, 100)
    assert run(F, params(), [far])[0] == 0
    far["octave"] = 1                                                        # 10 * 1.2 = 12: |dx| < r is strict
    F1 = frame(oracle, [(100, 100)], octaves=[1])
    assert run(F1, params(), [far])[0] == 0
    far["octave"] = 2                                                        # r = 14.4, levels 1..3
    assert run(F1, params(), [far])[0] == 1
    # a translated pose: the camera moved +1 along x, the point appears 50 px further left at z = 2
    T = EYE.copy()
    T[3] = -1.0
    assert run(frame(oracle, [(52, 100)]), params(), [at_pixel(102, 100)], Tcw=T)[0] == 1


def test_level_range_follows_the_motion(oracle):
    # last-frame octave 2; current features on levels 0..4 at growing distance
    F = frame(oracle, [(100, 100)] * 5, octaves=[0, 1, 2, 3, 4])
    for i in range(5):
        F["desc"][i, 0] = [0x1F, 0x07, 0x0F, 0x03, 0x01][i]                 # distances 5, 3, 4, 2, 1
    q = [at_pixel(100, 100, octave=2)]
    still, fwd, back = EYE, EYE.copy(), EYE.copy()
    fwd[11], back[11] = 0.5, -0.5                                            # tlc.z = +-0.5 against mb = 0.1 (:1385-1386)
    assert list(run(F, params(), q, Tlw=still)[2]) == [3]                    # levels 1..3: distance 2
    assert list(run(F, params(), q, Tlw=fwd)[2]) == [4]                      # levels >= 2: distance 1
    assert list(run(F, params(), q, Tlw=back)[2]) == [1]                     # levels 0..2: distance 3
    assert list(run(F, params(mono=1), q, Tlw=fwd)[2]) == [3]                # monocular: never forward/backward


def test_right_coordinate_gate(oracle):
    F = frame(oracle, [(100, 100), (103, 100)])
    F["desc"][1, 0] = 1
    # ur = u - mbf * invzc = 101 - 10 / 2 = 96 (:1446); feature 0 has a right coordinate far from it
    u_right = np.array([50.0, -1.0], np.float32)
    assert list(run(F, params(), [at_pixel(101, 100)], u_right=u_right)[2]) == [1]
    u_right[0] = 96.0 + 10.0                                                 # er == radius passes (:1448)
    assert list(run(F, params(), [at_pixel(101, 100)], u_right=u_right)[2]) == [0]


def test_rotation_histogram(oracle):
    # bin = round(rot / 30) (:1373, :1473): 12 matches with rot 0 (bin 0), one with rot 100 (bin 3), one with 350 (bin 12)
    pts = [(40 + 40 * i, 100) for i in range(14)]
    angles = [0.0] * 12 + [260.0, 10.0]
    F = frame(oracle, pts, angles=angles)
    qs = [at_pixel(40 + 40 * i, 100) for i in range(14)]
    n, match, qm = run(F, params(check=1), qs)
    # max1 = 12, max2 = max3 = 1 < 1.2: only the largest bin survives (:1670-1674)
    assert n == 12 and list(match) == list(range(12)) + [-1, -1] and list(qm) == list(range(14))
    n, match, _ = run(F, params(check=0), qs)
    assert n == 14 and list(match) == list(range(14))
    # with 10 in the largest bin, 1 >= 1.0 keeps the second and third bins
    F = frame(oracle, pts[:12], angles=[0.0] * 10 + [260.0, 10.0])
    n, match, _ = run(F, params(check=1), [at_pixel(40 + 40 * i, 100) for i in range(12)])
    assert n == 12 and list(match) == list(range(12))
    # negative differences wrap by 360 (:1471): last angle 0, current 350 -> rot 10 -> bin 0
    F = frame(oracle, pts[:13], angles=[350.0] * 12 + [200.0])
    n, match, _ = run(F, params(check=1), [at_pixel(40 + 40 * i, 100) for i in range(13)])
    assert n == 12 and match[12] == -1
