#!/usr/bin/env python3
"""Throughput of the input staging kernels on 512 KITTI-shape frames resident in HBM."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orb_slam2_amd as oa  # noqa: E402
from hip_rt import DeviceArray  # noqa: E402

w, h, n = 1241, 376, 512
stride = 1280
ex = oa.ORBextractor(2000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
rng = np.random.default_rng(0)
d_bgr = DeviceArray.from_host(rng.integers(0, 256, (n, h, w * 3), dtype=np.uint8))
d_gray = DeviceArray((n, h, stride), np.uint8, fill_byte=0)
d_rect = DeviceArray((n, h, stride), np.uint8, fill_byte=0)
ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
d_mx = DeviceArray.from_host((xs * np.float32(0.98) + np.float32(7.3) + np.float32(0.01) * ys).astype(np.float32))
d_my = DeviceArray.from_host((ys * np.float32(0.97) + np.float32(3.1)).astype(np.float32))


def timed(fn, reps=10):
    fn()
    ex.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ex.synchronize()
    return (time.perf_counter() - t0) / reps


tg = timed(lambda: ex.gray_batch_device(n, d_bgr.ptr, w, h, 3 * w, 3 * w * h, 3, 0, d_gray.ptr, stride, h * stride, sync=False))
tr = timed(lambda: ex.remap_batch_device(n, d_gray.ptr, w, h, stride, h * stride, d_mx.ptr, d_my.ptr, w, d_rect.ptr, w, h,
                                         stride, h * stride, sync=False))
px = n * w * h
print("gray  (BGR -> 8UC1): %.3f ms per %d frames, %.0f GB/s (4 B/px)" % (tg * 1e3, n, 4 * px / tg / 1e9))
print("remap (bilinear)   : %.3f ms per %d frames, %.0f GB/s (2 B/px + maps)" % (tr * 1e3, n, 2 * px / tr / 1e9))
