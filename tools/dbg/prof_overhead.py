"""Cost of the stage-timing events inside the timed region: the same 512-frame step with and without them."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import orb_slam2_amd as oa

run = bench.DeviceRun(torch, oa, bench.C3, 512, 0, first_idx=0)
for rep in range(3):
    for prof in (False, True):
        run.ex.set_profiling(prof)
        sec = bench.timed_steps(run, 200, 5)
        print("stage events", prof, "%.4f ms/step" % (sec * 1e3), flush=True)
run.close()
