#!/usr/bin/env python3
"""The reference's STEREO call pattern (src/Frame.cc:78-81): two extractor instances, each called from a
thread of its own that is created for the call and joined after it, then the stereo association.  Used under
rocprofv3 --kernel-trace to read off which hardware queues the two contexts' streams land on."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orb_slam2_amd as oa  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

W, H, NF = 1241, 376, 2000
PAIRS = int(sys.argv[1]) if len(sys.argv) > 1 else 50
left = [synth.make_frame(W, H, 3, 2 * i) for i in range(4)]
right = [np.ascontiguousarray(np.roll(f, -7, axis=1)) for f in left]
exl = oa.ORBextractor(NF, 1.2, 8, 20, 7, device=0, max_width=W, max_height=H, max_batch=1)
exr = oa.ORBextractor(NF, 1.2, 8, 20, 7, device=0, max_width=W, max_height=H, max_batch=1)
t = []
for p in range(PAIRS + 5):
    t0 = time.perf_counter()
    a = threading.Thread(target=exl, args=(left[p % 4],))
    b = threading.Thread(target=exr, args=(right[p % 4],))
    a.start(); b.start(); a.join(); b.join()
    exl.stereo_match(exr, 0.5371657, 386.1448)
    if p >= 5:
        t.append(time.perf_counter() - t0)
print("two contexts, two threads per pair + stereo association: median %.3f ms per pair over %d pairs" %
      (1e3 * float(np.median(t)), PAIRS))
exl.close()
exr.close()
