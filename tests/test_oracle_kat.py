"""Known-answer tests that pin the CPU oracle (oracle/orb_oracle.cpp).

The reference holds no tests or golden vectors for the extractor, so these are
hand-derivable facts plus independent brute-force restatements in numpy/Python of the
published OpenCV 3.4.5 algorithms.  None of this needs a GPU.
"""
import hashlib
import math
import struct

import numpy as np
import pytest

from orb_slam2_amd import synth


def test_umax_and_circle_area(oracle):
    p = oracle.params(1000, 1.2, 8)
    umax = list(p.umax)
    assert umax == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert (2 * umax[0] + 1) + 2 * sum(2 * u + 1 for u in umax[1:]) == 749


def test_pattern_table(oracle):
    pat = oracle.pattern().astype(np.int32)
    assert hashlib.sha256(struct.pack("<1024i", *pat)).hexdigest() == \
        "7e645581387b82784797e8adddb9b6f0c12611859fda09ca8a9bec96d767a05f"
    assert list(pat[:4]) == [8, -3, 9, 5] and list(pat[-4:]) == [-1, -6, 0, -11]
    assert pat.min() == -13 and pat.max() == 12
    r = np.hypot(pat[0::2], pat[1::2]).max()
    assert abs(r - 18.38) < 0.01


@pytest.mark.parametrize("nf,nl,expect", [
    (1000, 8, [217, 181, 151, 126, 105, 87, 73, 60]),
    (2000, 8, [434, 362, 302, 251, 209, 175, 145, 122]),
    (1200, 8, [261, 217, 181, 151, 126, 105, 87, 72]),
    (8000, 12, [1502, 1251, 1043, 869, 724, 604, 503, 419, 349, 291, 243, 202]),
])
def test_quotas(oracle, nf, nl, expect):
    p = oracle.params(nf, 1.2, nl)
    assert list(p.quota)[:nl] == expect


def test_scale_tables(oracle):
    p = oracle.params(8000, 1.2, 12)
    s = np.float32(1.0)
    sf = float(np.float32(1.2))  # the double member holds (double)1.2f
    for i in range(12):
        assert p.scale[i] == s
        assert p.sigma2[i] == np.float32(s * s)
        assert p.inv_scale[i] == np.float32(1.0) / s
        s = np.float32(float(s) * sf)


@pytest.mark.parametrize("w,h,nl,sizes", [
    (640, 480, 8, [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]),
    (1241, 376, 8, [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]),
    (752, 480, 8, [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)]),
    (3840, 2160, 12, [(3840, 2160), (3200, 1800), (2667, 1500), (2222, 1250), (1852, 1042), (1543, 868),
                      (1286, 723), (1072, 603), (893, 502), (744, 419), (620, 349), (517, 291)]),
])
def test_level_sizes(oracle, w, h, nl, sizes):
    p = oracle.params(1000, 1.2, nl)
    assert [oracle.level_size(p, w, h, l) for l in range(nl)] == sizes


def test_blur_constant(oracle):
    for c in (0, 1, 100, 200, 254, 255):
        img = np.full((40, 50), c, np.uint8)
        out = oracle.gaussian_blur7(img)
        assert (out == min(255, (c * 257 * 257 + 32768) >> 16)).all()
    assert min(255, (100 * 257 * 257 + 32768) >> 16) == 101


def test_blur_vs_numpy(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    pad = np.pad(img.astype(np.int64), 3, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    rows = sum(k[i] * pad[:, i:i + 53] for i in range(7))
    out = sum(k[j] * rows[j:j + 37, :] for j in range(7))
    out = np.minimum(255, (out + 32768) >> 16).astype(np.uint8)
    assert (oracle.gaussian_blur7(img) == out).all()


def test_resize_constant_and_identity(oracle):
    for c in (0, 7, 128, 255):
        img = np.full((48, 60), c, np.uint8)
        assert (oracle.resize_linear(img, 50, 40) == c).all()
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (31, 45), dtype=np.uint8)
    assert (oracle.resize_linear(img, 45, 31) == img).all()


def _resize_ref(src, dw, dh):
    """Independent numpy restatement of the OpenCV 3.4.5 8UC1 INTER_LINEAR fixed-point path."""
    sh, sw = src.shape
    f32 = np.float32

    def taps(dn, sn, clamp_x):
        inv = float(dn) / sn
        scale = 1.0 / inv
        f = ((np.arange(dn) + 0.5) * scale - 0.5).astype(f32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(f32)).astype(f32)
        if clamp_x:
            lo = s < 0
            f[lo] = 0
            s[lo] = 0
            hi = s >= sn - 1
            f[hi] = 0
            s[hi] = sn - 1
        c0 = np.rint((f32(1.0) - f) * f32(2048)).astype(np.int64)
        c1 = np.rint(f * f32(2048)).astype(np.int64)
        return s, c0, c1

    sx, a0, a1 = taps(dw, sw, True)
    sy, b0, b1 = taps(dh, sh, False)
    S = src.astype(np.int64)
    sx1 = np.minimum(sx + 1, sw - 1)
    T = S[:, sx] * a0[None, :] + S[:, sx1] * a1[None, :]
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    v = (((b0[:, None] * (T[y0] >> 4)) >> 16) + ((b1[:, None] * (T[y1] >> 4)) >> 16) + 2) >> 2
    return v.astype(np.uint8)


@pytest.mark.parametrize("sw,sh,dw,dh", [(120, 90, 100, 75), (1241, 376, 1034, 313), (179, 134, 149, 112),
                                         (64, 64, 53, 53), (100, 80, 130, 97)])
def test_resize_vs_numpy(oracle, sw, sh, dw, dh):
    rng = np.random.default_rng(sw * 7 + dh)
    img = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
    assert (oracle.resize_linear(img, dw, dh) == _resize_ref(img, dw, dh)).all()


RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _fast_bruteforce(img, t):
    """FAST-9/16 + strict 3x3 NMS straight from the definition (no early-outs, no cornerScore tricks)."""
    h, w = img.shape
    I = img.astype(int)
    score = np.zeros((h, w), int)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            v = I[y, x]
            d = [v - I[y + dy, x + dx] for dx, dy in RING]
            best = -10 ** 9
            for s in range(16):
                arc = [d[(s + k) % 16] for k in range(9)]
                best = max(best, min(arc), min(-a for a in arc))
            if best > t:  # nine contiguous all > t or all < -t
                score[y, x] = best - 1
                assert score[y, x] <= 255
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s == 0 and not (t == 0):
                continue
            nb = score[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (s > nb).all() and s > 0:
                out.append((x, y, s))
    return out


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
@pytest.mark.parametrize("t", [7, 20])
def test_fast_vs_bruteforce(oracle, seed, t):
    rng = np.random.default_rng(seed)
    if seed < 2:
        img = rng.integers(0, 256, (38, 37), dtype=np.uint8)
    else:
        img = synth.make_frame(96, 64, 9, seed)[10:48, 20:57]
    got = oracle.fast(img, t)
    exp = _fast_bruteforce(img, t)
    assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in got] == exp


def test_fast_known_scores(oracle):
    # an isolated bright pixel: all 16 ring pixels darker by 150 -> S = 150, score 149
    img = np.full((21, 21), 50, np.uint8)
    img[10, 10] = 200
    got = oracle.fast(img, 20)
    assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in got] == [(10, 10, 149)]
    # the corner of a bright square is a plateau of equal scores: strict NMS removes all of it
    img = np.full((21, 21), 50, np.uint8)
    img[10:, 10:] = 200
    assert len(oracle.fast(img, 20)) == 0 and _fast_bruteforce(img, 20) == []
    # a bright wedge whose tip is brighter than its body: tip survives with min |diff| - 1
    img[10:, 10:] = 180
    img[10, 10] = 230
    got = [(int(k["x"]), int(k["y"]), int(k["response"])) for k in oracle.fast(img, 20)]
    assert got == _fast_bruteforce(img, 20) and (10, 10, 179) in got


def test_fast_atan2_axes_and_accuracy(oracle):
    assert oracle.fast_atan2(0, 1) == 0.0
    assert oracle.fast_atan2(1, 0) == 90.0
    assert oracle.fast_atan2(0, -1) == 180.0
    assert oracle.fast_atan2(-1, 0) == 270.0
    assert oracle.fast_atan2(0, 0) == 0.0
    rng = np.random.default_rng(5)
    for _ in range(2000):
        y, x = rng.integers(-1248480, 1248481, 2)
        if x == 0 and y == 0:
            continue
        a = oracle.fast_atan2(float(y), float(x))
        t = math.degrees(math.atan2(y, x)) % 360.0
        d = abs(a - t)
        assert min(d, 360 - d) < 0.02  # the polynomial's documented accuracy


def _fast_atan2_f32(y, x):
    f = np.float32
    r2d = f(180 / math.pi)
    p1, p3, p5, p7 = (f(0.9997878412794807) * r2d, f(-0.3258083974640975) * r2d,
                      f(0.1555786518463281) * r2d, f(-0.04432655554792128) * r2d)
    eps = f(2.2204460492503131e-16)
    x, y = f(x), f(y)
    ax, ay = abs(x), abs(y)
    if ax >= ay:
        c = ay / (ax + eps)
        c2 = c * c
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    else:
        c = ax / (ay + eps)
        c2 = c * c
        a = f(90.0) - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    if x < 0:
        a = f(180.0) - a
    if y < 0:
        a = f(360.0) - a
    return float(a)


def test_fast_atan2_bits_vs_numpy_float32(oracle):
    rng = np.random.default_rng(6)
    for _ in range(3000):
        y, x = rng.integers(-1248480, 1248481, 2)
        assert oracle.fast_atan2(float(y), float(x)) == _fast_atan2_f32(y, x)


def test_ic_angle_ramps(oracle):
    umax = list(oracle.params(1000, 1.2, 8).umax)
    yy, xx = np.mgrid[0:41, 0:41]
    right = (100 + 3 * (xx - 20)).astype(np.uint8)
    down = (100 + 3 * (yy - 20)).astype(np.uint8)
    assert oracle.ic_angle(right, 20, 20, umax) == 0.0
    assert oracle.ic_angle(down, 20, 20, umax) == 90.0
    assert oracle.ic_angle(right[:, ::-1].copy(), 20, 20, umax) == 180.0
    assert oracle.ic_angle(down[::-1].copy(), 20, 20, umax) == 270.0
    # brute-force moments on a random patch
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (41, 41), dtype=np.uint8)
    m10 = m01 = 0
    for v in range(-15, 16):
        d = umax[abs(v)]
        for u in range(-d, d + 1):
            m10 += u * int(img[20 + v, 20 + u])
            m01 += v * int(img[20 + v, 20 + u])
    assert oracle.ic_angle(img, 20, 20, umax) == oracle.fast_atan2(float(m01), float(m10))


def test_sincos_correctly_rounded(oracle):
    """D6: a, b are the float32 roundings of cos/sin of the float32 radian angle."""
    factor = np.float32(math.pi / 180.0)
    rng = np.random.default_rng(8)
    angles = np.concatenate([np.array([0, 45, 90, 135, 180, 225, 270, 315, 359.99997, 360], np.float32),
                             rng.uniform(0, 360, 200000).astype(np.float32)])
    for a in angles[:20000]:
        c, s = oracle.sincos_deg(float(a))
        r = float(np.float32(a) * factor)
        assert c == float(np.float32(math.cos(r))) and s == float(np.float32(math.sin(r))), a


def test_descriptor_ramp_angle0(oracle):
    """On I = 4x + c with angle 0 (a=1, b=0) bit k is [x(p_2k) < x(p_2k+1)]."""
    yy, xx = np.mgrid[0:61, 0:61]
    img = (8 + 4 * xx).astype(np.uint8)  # 8..248, un-clipped
    blurred = oracle.gaussian_blur7(img)
    pat = oracle.pattern().astype(int).reshape(256, 4)
    exp = np.zeros(32, np.uint8)
    for k in range(256):
        if pat[k, 0] < pat[k, 2]:
            exp[k // 8] |= 1 << (k % 8)
    assert (oracle.descriptor(blurred, 30, 30, 0.0) == exp).all()
    # rotating the ramp by 180 degrees and the keypoint by 180 gives the same bits
    assert (oracle.descriptor(blurred[:, ::-1].copy(), 30, 30, 180.0) == exp).all()


def test_descriptor_vs_python(oracle):
    rng = np.random.default_rng(9)
    img = oracle.gaussian_blur7(rng.integers(0, 256, (61, 61), dtype=np.uint8))
    pat = oracle.pattern().astype(int).reshape(512, 2)
    f = np.float32
    for ang in (0.0, 33.3, 90.0, 123.456, 271.0, 359.9):
        a, b = oracle.sincos_deg(ang)
        a, b = f(a), f(b)
        vals = []
        for x, y in pat:
            # fmaf evaluated in double: the products of a float32 by a small integer are exact in double
            row = float(f(float(x) * float(b) + float(f(f(y) * a))))
            col = float(f(float(x) * float(a) - float(f(f(y) * b))))
            vals.append(int(img[30 + int(np.rint(row)), 30 + int(np.rint(col))]))
        exp = np.zeros(32, np.uint8)
        for k in range(256):
            if vals[2 * k] < vals[2 * k + 1]:
                exp[k // 8] |= 1 << (k % 8)
        assert (oracle.descriptor(img, 30, 30, ang) == exp).all()


def test_measurement_variants_of_the_pinned_decisions(oracle):
    """The hooks tools/tolerance_study.py measures with: the descriptor variant (0, 0) IS the pinned descriptor,
    the other variants differ from it in few bits at most, and another quadtree tie order keeps the per-level
    counts inside [N, N + 3] and is switched off again by mode 0."""
    import ctypes as C
    from orb_slam2_amd import synth
    L = oracle.lib()
    L.orc_descriptor_variant.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
    L.orc_set_d2_mode.argtypes = [C.c_int]
    L.orc_get_d2_sensitive.argtypes = [C.c_void_p, C.c_int]
    img = synth.make_frame(640, 480, 1, 2)
    orc = oracle.Oracle(1000, 1.2, 8, 20, 7)
    kps, desc = orc.extract(img)
    b0 = np.ascontiguousarray(orc.blurred(0))
    tmp = np.zeros(32, np.uint8)
    lvl0 = np.nonzero(kps["octave"] == 0)[0][:100]
    for i in lvl0:
        x, y, ang = int(kps["x"][i]), int(kps["y"][i]), float(kps["angle"][i])
        L.orc_descriptor_variant(b0.ctypes.data, b0.shape[1], x, y, ang, 0, 0, tmp.ctypes.data)
        assert (tmp == desc[i]).all()
        for rot, sc in ((1, 0), (0, 1), (0, 2), (0, 5)):
            L.orc_descriptor_variant(b0.ctypes.data, b0.shape[1], x, y, ang, rot, sc, tmp.ctypes.data)
            assert np.unpackbits(tmp ^ desc[i]).sum() <= 4
    quota = orc.params.quota
    try:
        for mode in (1, 2):
            L.orc_set_d2_mode(mode)
            k2, _ = orc.extract(img)
            for l in range(8):
                n = int((k2["octave"] == l).sum())
                assert n == int((kps["octave"] == l).sum()) or quota[l] <= n <= quota[l] + 3
    finally:
        L.orc_set_d2_mode(0)
    k3, d3 = orc.extract(img)
    assert (k3 == kps).all() and (d3 == desc).all()
    assert all(L.orc_get_d2_sensitive(orc._c, l) in (0, 1) for l in range(8))
