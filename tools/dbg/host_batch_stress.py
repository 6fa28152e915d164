#!/usr/bin/env python3
"""Stress of the pipelined host batch path: many calls with changing batch sizes and chunk sizes, two contexts on two
threads, every result compared with the first call's for the same frames.  Run on the GPU box."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import orb_slam2_amd as oa  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

W, H, NF, NMAX = 752, 480, 1200, 160
base = synth.make_batch(32, W, H, 4, 0)
failures = []


def worker(tid, seconds):
    rng = np.random.default_rng(tid)
    ex = oa.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=NMAX, keep_pyramid=False)
    cap = ex.max_keypoints(W, H)
    ref = {}
    calls = 0
    t_end = time.time() + seconds
    while time.time() < t_end:
        n = int(rng.integers(16, NMAX + 1))
        ex.set_debug_flag(oa.FLAG_PIPE_CHUNK, int(rng.choice([0, 0, 7, 16, 33, 64])))
        order = rng.integers(0, len(base), n)
        frames = [np.ascontiguousarray(base[i]) for i in order]
        kps, desc, cnt = np.zeros((n, cap), oa.KP_DTYPE), np.zeros((n, cap, 32), np.uint8), np.zeros(n, np.int32)
        ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        strides = (C.c_int * n)(*[f.strides[0] for f in frames])
        rc = ex._L.orbx_extract_batch(ex._ctx, n, ptrs, W, H, strides, kps.ctypes.data_as(C.c_void_p), cap,
                                      desc.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p))
        if rc != 0:
            failures.append("thread %d: rc %d" % (tid, rc))
            break
        for j, i in enumerate(order):
            k = int(cnt[j])
            got = (k, kps[j, :k].tobytes(), desc[j, :k].tobytes())
            if int(i) not in ref:
                ref[int(i)] = got
            elif ref[int(i)] != got:
                failures.append("thread %d: frame %d differs in call %d" % (tid, i, calls))
        calls += 1
    ex.close()
    print("thread %d: %d calls, %d distinct frames" % (tid, calls, len(ref)))


threads = [threading.Thread(target=worker, args=(t, float(sys.argv[1]) if len(sys.argv) > 1 else 15.0)) for t in range(2)]
for t in threads:
    t.start()
for t in threads:
    t.join()
print("failures:", failures[:5] if failures else "none")
sys.exit(1 if failures else 0)
