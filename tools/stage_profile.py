#!/usr/bin/env python3
"""Stage times (kernels alone on the GPU, stages serialised) and the overlapped step time of one configuration.
usage: stage_profile.py width height nfeatures nlevels batch"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import orb_slam2_amd as oa  # noqa: E402
from hip_rt import DeviceArray  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

w, h, nf, nl, n = (int(v) for v in sys.argv[1:6])
stride = (w + 63) // 64 * 64
base = synth.make_batch(min(n, 4), w, h, 5, 0)
host = np.zeros((n, h, stride), np.uint8)
for i in range(n):
    host[i, :, :w] = base[i % len(base)] if (i // len(base)) % 2 == 0 else 255 - base[i % len(base)]
frames = DeviceArray.from_host(host)
ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
cap = ex.max_keypoints(w, h)
d_kps, d_desc, d_n = DeviceArray((n, cap, 28), np.uint8), DeviceArray((n, cap, 32), np.uint8), DeviceArray((n,), np.int32, fill_byte=0)


def step():
    ex.extract_batch_device(frames.ptr, n, w, h, stride, h * stride, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=False)


for _ in range(3):
    step()
ex.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
ex.synchronize()
print("%dx%d nf=%d levels=%d batch=%d: overlapped step %.3f ms" % (w, h, nf, nl, n, (time.perf_counter() - t0) * 100))
ex.set_debug_flag(oa.FLAG_NO_OVERLAP, 1)
ex.set_profiling(True)
acc = {}
step()
ex.synchronize()
for _ in range(5):
    step()
    ex.synchronize()
    for k, v in ex.stage_ms().items():
        acc[k] = acc.get(k, 0.0) + v / 5
print("stages alone (ms):", {k: round(v, 3) for k, v in acc.items()}, "sum %.3f" % sum(acc.values()))
print("keypoints/frame %.1f" % d_n.to_host().mean())
