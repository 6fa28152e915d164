"""Headline configuration (KITTI-shape frames resident in HBM) against the batch size (run on the GPU box):
     gpurun -- 'python tools/c3_batch.py 32 64 128 256 512 1024'"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import orb_slam2_amd as oa  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [512]:
    run = bench.DeviceRun(torch, oa, bench.C3, n, 0, first_idx=0, n_base=min(n, 64))
    sec = bench.timed_steps(run, max(20, 51200 // n), 5)
    print("batch %4d: %.4f ms per step, %.0f frames/s" % (n, 1e3 * sec, n / sec), flush=True)
    run.close()
    del run
    torch.cuda.empty_cache()
