#!/usr/bin/env python3
"""Latency of orbx_extract itself (no Python-side allocation in the loop), with per-stage device times."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orb_slam2_amd as oa  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

w, h, nf = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (640, 480, 1000)
ex = oa.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, keep_pyramid=False)
L = ex._L
img = synth.make_frame(w, h, 1, 0)
cap = ex.max_keypoints(w, h)
kps = np.zeros(cap, oa.KP_DTYPE)
desc = np.zeros((cap, 32), np.uint8)
n = C.c_int(0)
args = (ex._ctx, img.ctypes.data_as(C.c_void_p), w, h, img.strides[0], kps.ctypes.data_as(C.c_void_p), cap,
        desc.ctypes.data_as(C.c_void_p), C.byref(n))
reps = int(sys.argv[4]) if len(sys.argv) >= 5 else 1000
for name, no_graph in (("one HIP graph launch per call", 0), ("launches issued one by one", 1)):
    ex.set_debug_flag(oa.FLAG_NO_GRAPH, no_graph)
    for _ in range(20):
        L.orbx_extract(*args)
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        L.orbx_extract(*args)
        t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e3
    print("orbx_extract %dx%d, %s: median %.3f ms p10 %.3f p90 %.3f, %d keypoints (%d calls)"
          % (w, h, name, np.median(t), np.percentile(t, 10), np.percentile(t, 90), n.value, reps))
ex.set_profiling(True)
L.orbx_extract(*args)
print("device stage ms:", ex.stage_ms())
