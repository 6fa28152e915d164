"""Why the 4K figure of the default bench line differs from `bench.py --config c5`: the same DeviceRun timed with
and without stage events, and with the default number of base frames."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import orb_slam2_amd as oa

for n_base, prof in ((2, False), (2, True), (None, False), (None, True)):
    run = bench.DeviceRun(torch, oa, bench.C5, 128, 0, first_idx=2000, n_base=n_base)
    run.ex.set_profiling(prof)
    sec = bench.timed_steps(run, 20, 3)
    print("n_base", n_base, "stage events", prof, "%.3f ms/step" % (sec * 1e3), "kp/frame", float(run.d_n.float().mean()))
    run.close()
    del run
