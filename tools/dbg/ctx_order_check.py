"""Does a context created after another one was closed run its batches as fast?  (Hardware queues are handed to
streams by the runtime; a later context may get another mapping.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import orb_slam2_amd as oa

cfgs = {"c3": (bench.C3, 512, 200), "c5": (bench.C5, 128, 20)}
for name in sys.argv[1:] or ["c3", "c3", "c3", "c5", "c5", "c3"]:
    cfg, n, steps = cfgs[name]
    run = bench.DeviceRun(torch, oa, cfg, n, 0, first_idx=0)
    sec = bench.timed_steps(run, steps, 3)
    print(name, "%.3f ms/step" % (sec * 1e3), flush=True)
    run.close()
    del run
