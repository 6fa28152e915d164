"""GPU parity of Frame::UndistortKeyPoints + AssignFeaturesToGrid (reference src/Frame.cc:421-451,
232-247) against the CPU oracle, bit-exact, on the keypoints of real extractions."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth
from hip_rt import DeviceArray

pytestmark = pytest.mark.gpu

K = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)            # TUM1.yaml
DIST = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)


@pytest.mark.parametrize("dist", [DIST, np.zeros(5, np.float32), np.array([-0.28, 0.07, 1e-4, -2e-4, 0], np.float32)])
def test_undistort_and_grid(oracle, dist):
    w, h, nf, nl, n = 640, 480, 1000, 8, 3
    imgs = synth.make_batch(n, w, h, 1, 70)
    stride = 640
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    d_img = DeviceArray.from_host(imgs)
    d_kps, d_desc = DeviceArray((n, cap, 28), np.uint8, fill_byte=0), DeviceArray((n, cap, 32), np.uint8)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    d_un = DeviceArray((n, cap, 28), np.uint8, fill_byte=0)
    d_cs = DeviceArray((n, 64 * 48 + 1), np.int32, fill_byte=0xFF)
    d_ci = DeviceArray((n, cap), np.int32, fill_byte=0xFF)
    # image bounds as Frame::ComputeImageBounds derives them (src/Frame.cc:452-482)
    if dist[0] != 0:
        c = oracle.undistort_points(np.array([[0, 0], [w, 0], [0, h], [w, h]], np.float32), K, dist)
        min_x, max_x = min(c[0, 0], c[2, 0]), max(c[1, 0], c[3, 0])
        min_y, max_y = min(c[0, 1], c[1, 1]), max(c[2, 1], c[3, 1])
    else:
        min_x, max_x, min_y, max_y = np.float32(0), np.float32(w), np.float32(0), np.float32(h)
    inv_w = np.float32(64) / np.float32(max_x - min_x)
    inv_h = np.float32(48) / np.float32(max_y - min_y)
    ex.extract_batch_device(d_img.ptr, n, w, h, stride, h * stride, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=False)
    ex.undistort_grid_batch_device(n, d_kps.ptr, cap, d_n.ptr, K, dist, float(min_x), float(min_y), float(inv_w),
                                   float(inv_h), d_un.ptr, d_cs.ptr, d_ci.ptr, sync=True)
    counts = d_n.to_host()
    kps = d_kps.to_host().reshape(n, cap * 28).view(oa.KP_DTYPE).reshape(n, cap)
    un = d_un.to_host().reshape(n, cap * 28).view(oa.KP_DTYPE).reshape(n, cap)
    cs, ci = d_cs.to_host(), d_ci.to_host()
    for f in range(n):
        k = counts[f]
        xy = np.stack([kps[f, :k]["x"], kps[f, :k]["y"]], 1)
        oxy = oracle.undistort_points(xy, K, dist)
        assert (un[f, :k]["x"].view(np.uint32) == oxy[:, 0].view(np.uint32)).all()
        assert (un[f, :k]["y"].view(np.uint32) == oxy[:, 1].view(np.uint32)).all()
        for name in ("size", "angle", "response", "octave", "class_id"):
            assert (un[f, :k][name] == kps[f, :k][name]).all()
        ostart, oidx = oracle.assign_grid(oxy, min_x, min_y, inv_w, inv_h)
        assert (cs[f] == ostart).all()
        assert (ci[f, :ostart[-1]] == oidx).all()
        assert ostart[-1] > 0.9 * k
    ex.close()
    for a in (d_img, d_kps, d_desc, d_n, d_un, d_cs, d_ci):
        a.free()
