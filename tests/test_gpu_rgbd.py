"""GPU parity of Frame::ComputeStereoFromRGBD (reference src/Frame.cc:662-682) with the depth conversion of
Tracking::GrabImageRGBD (src/Tracking.cc:230-231) against the CPU oracle: mvuRight and mvDepth bit for bit, on 16-bit
and float32 depth maps with holes, for a batch of frames."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth

pytestmark = pytest.mark.gpu

K = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)
DIST = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)


@pytest.mark.parametrize("is_f32,factor,pad", [(False, np.float32(1.0) / np.float32(5000.0), 0), (False, 1.0, 6),
                                                (True, 1.0, 0), (True, 0.25, 10)])
def test_rgbd_depth_after_extraction(oracle, is_f32, factor, pad):
    from hip_rt import DeviceArray
    w, h, n = 640, 480, 3
    rng = np.random.default_rng(7 + pad)
    frames = np.stack([synth.make_frame(w, h, 2, 30 + i) for i in range(n)])
    if is_f32:
        depth = rng.uniform(0.3, 8.0, (n, h, w + pad)).astype(np.float32)
        depth[rng.random(depth.shape) < 0.2] = 0.0            # holes
        depth[rng.random(depth.shape) < 0.02] = -1.0          # and invalid negative samples
    else:
        depth = rng.integers(0, 40000, (n, h, w + pad)).astype(np.uint16)
        depth[rng.random(depth.shape) < 0.2] = 0
    ex = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    d_img, d_depth = DeviceArray.from_host(frames), DeviceArray.from_host(depth)
    d_kps, d_un = DeviceArray((n, cap), oa.KP_DTYPE), DeviceArray((n, cap), oa.KP_DTYPE)
    d_desc = DeviceArray((n, cap, 32), np.uint8)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    d_cs, d_ci = DeviceArray((n, 64 * 48 + 1), np.int32), DeviceArray((n, cap), np.int32)
    d_ur, d_z = DeviceArray((n, cap), np.float32, fill_byte=0x55), DeviceArray((n, cap), np.float32, fill_byte=0x55)
    ex.extract_batch_device(d_img.ptr, n, w, h, w, w * h, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=False)
    ex.undistort_grid_batch_device(n, d_kps.ptr, cap, d_n.ptr, K, DIST, 0.0, 0.0, 0.1, 0.1, d_un.ptr, d_cs.ptr, d_ci.ptr, sync=False)
    item = depth.dtype.itemsize
    ex.rgbd_depth_batch_device(n, d_depth.ptr, is_f32, w, h, (w + pad) * item, (w + pad) * h * item, float(factor), d_kps.ptr,
                               d_un.ptr, cap, d_n.ptr, 40.0, d_ur.ptr, d_z.ptr, sync=True)
    counts, kps, un = d_n.to_host(), d_kps.to_host(), d_un.to_host()
    ur, z = d_ur.to_host(), d_z.to_host()
    with_depth = 0
    for f in range(n):
        k = counts[f]
        xy = np.stack([kps[f, :k]["x"], kps[f, :k]["y"]], 1)
        eur, ez = oracle.stereo_from_rgbd(depth[f], float(factor), xy, un[f, :k]["x"], 40.0)
        assert (ur[f, :k].view(np.uint32) == eur.view(np.uint32)).all(), "mvuRight of frame %d" % f
        assert (z[f, :k].view(np.uint32) == ez.view(np.uint32)).all(), "mvDepth of frame %d" % f
        assert (ur[f, k:] == -1).all() and (z[f, k:] == -1).all()
        with_depth += int((ez > 0).sum())
    assert with_depth > 1000
    ex.close()
    for a in (d_img, d_depth, d_kps, d_un, d_desc, d_n, d_cs, d_ci, d_ur, d_z):
        a.free()


def test_rgbd_depth_arguments():
    ex = oa.ORBextractor(500, 1.2, 4, 20, 7, max_width=320, max_height=240, max_batch=1, keep_pyramid=False)
    with pytest.raises(oa.OrbxError):
        ex.rgbd_depth_batch_device(1, 0, False, 320, 240, 640, 640 * 240, 1.0, 1, 1, 100, 1, 40.0, 1, 1)     # null depth map
    with pytest.raises(oa.OrbxError):
        ex.rgbd_depth_batch_device(1, 256, False, 320, 240, 600, 600 * 240, 1.0, 1, 1, 100, 1, 40.0, 1, 1)   # stride < width * 2
    with pytest.raises(oa.OrbxError):
        ex.rgbd_depth_batch_device(1, 258, True, 320, 240, 1280, 1280 * 240, 1.0, 1, 1, 100, 1, 40.0, 1, 1)  # float map not 4-byte aligned
    ex.close()
