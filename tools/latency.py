#!/usr/bin/env python3
"""Single-frame latency of the host-memory entry point (upload + whole path + download), the way
the reference's Tracking thread calls the extractor: one frame per call.
usage: python tools/latency.py [width height nfeatures [reps]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orb_slam2_amd as oa  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

w, h, nf = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (640, 480, 1000)
reps = int(sys.argv[4]) if len(sys.argv) >= 5 else 1000
imgs = [synth.make_frame(w, h, 1, i) for i in range(8)]
for keep in (True, False):
    ex = oa.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, keep_pyramid=keep)
    for i in range(20):
        ex(imgs[i % 8])
    t = []
    for i in range(reps):
        t0 = time.perf_counter()
        ex(imgs[i % 8])
        t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e3
    print("%dx%d nf=%d keep_pyramid=%s: median %.3f ms  p10 %.3f  p90 %.3f  (%d calls)"
          % (w, h, nf, keep, np.median(t), np.percentile(t, 10), np.percentile(t, 90), reps))
    ex.close()
