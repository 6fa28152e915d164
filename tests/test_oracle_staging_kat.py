"""Known-answer tests of the CPU restatement of the input staging stage: cv::cvtColor *2GRAY
(reference src/Tracking.cc:175-199) and cv::remap INTER_LINEAR with float maps (reference
Examples/Stereo/stereo_euroc.cc:137-138), OpenCV 3.4.5 8-bit arithmetic.  Expected values are
computed by hand from the fixed-point formulas."""
import numpy as np


def test_gray_of_pure_channels_and_neutral(oracle):
    px = np.zeros((1, 4, 3), np.uint8)
    px[0, 0] = (255, 0, 0)
    px[0, 1] = (0, 255, 0)
    px[0, 2] = (0, 0, 255)
    px[0, 3] = (200, 200, 200)
    # BGR order: B*1868, G*9617, R*4899, +8192, >>14
    assert oracle.cvt_gray(px, 0).tolist() == [[(255 * 1868 + 8192) >> 14, (255 * 9617 + 8192) >> 14,
                                                (255 * 4899 + 8192) >> 14, 200]]
    assert oracle.cvt_gray(px, 0).tolist() == [[29, 150, 76, 200]]
    # RGB order swaps the outer coefficients
    assert oracle.cvt_gray(px, 2).tolist() == [[76, 150, 29, 200]]
    rgba = np.concatenate([px, np.full((1, 4, 1), 9, np.uint8)], axis=2)
    assert (oracle.cvt_gray(rgba, 2) == oracle.cvt_gray(px, 2)).all()   # alpha ignored


def test_gray_matches_numpy_formula(oracle):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    b, g, r = (img[..., k].astype(np.int64) for k in range(3))
    exp = ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)
    assert (oracle.cvt_gray(img, 0) == exp).all()
    assert (oracle.cvt_gray(img[..., ::-1], 2) == exp).all()


def test_remap_identity_shift_and_border(oracle):
    rng = np.random.default_rng(6)
    src = rng.integers(0, 256, (40, 60), dtype=np.uint8)
    ys, xs = np.mgrid[0:40, 0:60].astype(np.float32)
    ident = oracle.remap_linear(src, xs, ys)
    # identity: the last row and column take the border branch with weight 32768 on the pixel itself
    assert (ident == src).all()
    # half a pixel to the right: (a*16384 + b*16384 + 16384) >> 15 = (a + b + 1) >> 1; the last column mixes with 0
    half = oracle.remap_linear(src, xs + np.float32(0.5), ys)
    a, b = src[:, :-1].astype(np.int32), src[:, 1:].astype(np.int32)
    assert (half[:, :-1] == ((a + b + 1) >> 1)).all()
    assert (half[:, -1] == ((src[:, -1].astype(np.int32) + 1) >> 1)).all()
    # coordinates are rounded to 1/32 px, ties to even: 1/64 -> 0.5/32 -> 0 ; 3/64 -> 1.5/32 -> 2/32
    t0 = oracle.remap_linear(src, xs + np.float32(1 / 64), ys)
    assert (t0 == src).all()
    t1 = oracle.remap_linear(src, xs + np.float32(3 / 64), ys)
    exp = ((a * (30 * 32 * 32) + b * (2 * 32 * 32) + 16384) >> 15)
    assert (t1[:, :-1] == exp).all()
    # entirely outside -> 0 ; one pixel outside on the left -> only the right tap counts
    out = oracle.remap_linear(src, xs - np.float32(100), ys)
    assert (out == 0).all()
    left = oracle.remap_linear(src, xs - np.float32(0.5), ys)
    assert (left[:, 0] == ((src[:, 0].astype(np.int32) + 1) >> 1)).all()
    # a constant image stays constant wherever all four taps are inside
    c = np.full((40, 60), 137, np.uint8)
    w = oracle.remap_linear(c, xs * np.float32(0.9) + np.float32(1.3), ys * np.float32(0.8) + np.float32(2.1))
    assert (w == 137).all()
