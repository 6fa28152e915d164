"""One 512-frame batch at a time (synchronised after every call): the free-running order against ORBX_FLAG_ORDERED."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
import orb_slam2_amd as oa

for cfg, n in ((bench.C3, 512), (bench.C3, 64), (bench.C2, 16)):
    run = bench.DeviceRun(torch, oa, cfg, n, 0, first_idx=0)
    for ordered in (0, 1, 0, 1):
        run.ex.set_debug_flag(oa.FLAG_ORDERED, ordered)
        t = []
        for i in range(60):
            run.ex.synchronize()
            t0 = time.perf_counter()
            run.step()
            run.ex.synchronize()
            t.append(time.perf_counter() - t0)
        print(cfg["name"], n, "frames, ordered", ordered, "median %.4f ms" % (1e3 * float(np.median(t[10:]))), flush=True)
    run.close()
    del run
