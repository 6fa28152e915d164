"""GPU parity of the input staging stage (gray conversion, rectification remap) against the CPU
oracle, bit-exact, through the C ABI's device-pointer entry points; and the staged frames feeding
the extractor on the same stream."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth
from hip_rt import DeviceArray

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("channels,blue_idx,w,h", [(3, 0, 752, 480), (3, 2, 641, 37), (4, 0, 322, 50), (4, 2, 1241, 376)])
def test_gray_batch(oracle, channels, blue_idx, w, h):
    rng = np.random.default_rng(channels * 10 + blue_idx)
    n = 3
    stride = w * channels + 5                      # odd source stride
    src = rng.integers(0, 256, (n, h, stride), dtype=np.uint8)
    dstride = (w + 63) // 64 * 64
    ex = oa.ORBextractor(500, 1.2, 4, 20, 7, max_width=max(w, 320), max_height=max(h, 240))
    d_src = DeviceArray.from_host(src)
    d_dst = DeviceArray((n, h, dstride), np.uint8, fill_byte=0)
    ex.gray_batch_device(n, d_src.ptr, w, h, stride, h * stride, channels, blue_idx, d_dst.ptr, dstride, h * dstride)
    got = d_dst.to_host()
    for i in range(n):
        img = src[i, :, :w * channels].reshape(h, w, channels)
        assert (got[i, :, :w] == oracle.cvt_gray(img, blue_idx)).all()
    assert (got[:, :, w:] == 0).all()              # nothing written beyond the row
    ex.close()
    d_src.free()
    d_dst.free()


def rectify_like_maps(w, h, k1=-0.28, shift=(3.7, -2.2), rot=0.01):
    """Maps shaped like cv::initUndistortRectifyMap output: radial distortion + small rotation + shift;
    some coordinates leave the image."""
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy, f = w / 2 - 0.5, h / 2 - 0.5, 0.8 * w
    x, y = (xs - cx) / f, (ys - cy) / f
    xr, yr = x * np.cos(rot) - y * np.sin(rot), x * np.sin(rot) + y * np.cos(rot)
    r2 = xr * xr + yr * yr
    d = 1 + k1 * r2 + 0.07 * r2 * r2
    return (xr * d * f + cx + shift[0]).astype(np.float32), (yr * d * f + cy + shift[1]).astype(np.float32)


@pytest.mark.parametrize("w,h", [(752, 480), (333, 201)])
def test_remap_batch(oracle, w, h):
    n = 2
    frames = synth.make_batch(n, w, h, 4, 50)
    sstride = w + 3
    src = np.zeros((n, h, sstride), np.uint8)
    src[:, :, :w] = frames
    mapx, mapy = rectify_like_maps(w, h)
    mapx[5:9, 10:40] = -7.3                         # a patch entirely outside
    mapx[20, :] = np.float32(w) - np.float32(0.75)  # straddling the right edge
    mapy[:, 50] = -0.5                              # straddling the top edge
    dstride = (w + 63) // 64 * 64
    ex = oa.ORBextractor(500, 1.2, 4, 20, 7, max_width=max(w, 320), max_height=max(h, 240))
    d_src, d_mx, d_my = DeviceArray.from_host(src), DeviceArray.from_host(mapx), DeviceArray.from_host(mapy)
    d_dst = DeviceArray((n, h, dstride), np.uint8, fill_byte=0x55)
    ex.remap_batch_device(n, d_src.ptr, w, h, sstride, h * sstride, d_mx.ptr, d_my.ptr, w, d_dst.ptr, w, h, dstride,
                          h * dstride)
    got = d_dst.to_host()
    for i in range(n):
        assert (got[i, :, :w] == oracle.remap_linear(src[i, :, :w], mapx, mapy)).all()
    ex.close()
    for a in (d_src, d_mx, d_my, d_dst):
        a.free()


def test_staged_frames_feed_the_extractor(oracle):
    """colour frames -> gray -> rectified -> extractor, all enqueued on one context without host round trips."""
    w, h, nf, nl, n = 752, 480, 1200, 8, 2
    gray = synth.make_batch(n, w, h, 4, 60)
    rng = np.random.default_rng(9)
    bgr = np.stack([np.clip(gray.astype(np.int32) + rng.integers(-20, 21, gray.shape), 0, 255) for _ in range(3)], -1)
    bgr = np.ascontiguousarray(bgr.astype(np.uint8))
    mapx, mapy = rectify_like_maps(w, h, k1=-0.1, shift=(1.25, 0.5), rot=0.002)
    stride = (w + 63) // 64 * 64
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    d_bgr, d_mx, d_my = DeviceArray.from_host(bgr), DeviceArray.from_host(mapx), DeviceArray.from_host(mapy)
    d_gray = DeviceArray((n, h, stride), np.uint8, fill_byte=0)
    d_rect = DeviceArray((n, h, stride), np.uint8, fill_byte=0)
    d_kps, d_desc = DeviceArray((n, cap, 28), np.uint8), DeviceArray((n, cap, 32), np.uint8)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    ex.gray_batch_device(n, d_bgr.ptr, w, h, 3 * w, 3 * w * h, 3, 0, d_gray.ptr, stride, h * stride, sync=False)
    ex.remap_batch_device(n, d_gray.ptr, w, h, stride, h * stride, d_mx.ptr, d_my.ptr, w, d_rect.ptr, w, h, stride,
                          h * stride, sync=False)
    ex.extract_batch_device(d_rect.ptr, n, w, h, stride, h * stride, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=True)
    counts, kps, desc = d_n.to_host(), d_kps.to_host(), d_desc.to_host()
    orc = oracle.Oracle(nf, 1.2, nl, 20, 7)
    for i in range(n):
        rect = oracle.remap_linear(oracle.cvt_gray(bgr[i], 0), mapx, mapy)
        okps, odesc = orc.extract(rect)
        k = counts[i]
        assert k == len(okps) and k > nf // 2
        got = np.ascontiguousarray(kps[i, :k]).reshape(-1).view(oa.KP_DTYPE)
        for f in oa.KP_DTYPE.names:
            assert (got[f] == okps[f]).all(), f
        assert (desc[i, :k] == odesc).all()
    ex.close()
    for a in (d_bgr, d_mx, d_my, d_gray, d_rect, d_kps, d_desc, d_n):
        a.free()
