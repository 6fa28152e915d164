#!/usr/bin/env python3
"""Throughput of the host-memory batch entry point (PCIe-inclusive): frames in pageable host memory
in, keypoints and descriptors in host memory out."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orb_slam2_amd as oa  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

w, h, nf, n = 1241, 376, 2000, int(sys.argv[1]) if len(sys.argv) > 1 else 512
base = synth.make_batch(16, w, h, 3, 0)
frames = [np.ascontiguousarray(base[i % 16]) for i in range(n)]
ex = oa.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
L = ex._L
cap = ex.max_keypoints(w, h)
kps = np.zeros((n, cap), oa.KP_DTYPE)
desc = np.zeros((n, cap, 32), np.uint8)
cnt = np.zeros(n, np.int32)
ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
strides = (C.c_int * n)(*[f.strides[0] for f in frames])
args = (ex._ctx, n, ptrs, w, h, strides, kps.ctypes.data_as(C.c_void_p), cap, desc.ctypes.data_as(C.c_void_p),
        cnt.ctypes.data_as(C.c_void_p))
assert L.orbx_extract_batch(*args) == 0
t = []
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for k in range(calls):
    sys.stderr.write("== call %d\n" % k)
    sys.stderr.flush()
    t0 = time.perf_counter()
    assert L.orbx_extract_batch(*args) == 0
    t.append(time.perf_counter() - t0)
best = min(t)
print("calls (ms): " + " ".join("%.2f" % (1e3 * v) for v in t))
print("orbx_extract_batch, %d KITTI-shape frames from/to host memory: %.2f ms -> %.0f frames/s (%.1f GB/s in)"
      % (n, best * 1e3, n / best, n * w * h / best / 1e9))
