"""C5 (3840x2160, 8000 features, 12 levels) rate against the batch size (run on the GPU box through gpurun):
     gpurun -- 'python tools/c5_batch.py 16 32 64 128'"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import orb_slam2_amd as oa  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [32]:
    run = bench.DeviceRun(torch, oa, bench.C5, n, 0, first_idx=2000, n_base=2)
    sec = bench.timed_steps(run, max(2, 320 // n), 2)
    st = None
    print("batch %4d: %.3f ms per step, %.0f frames/s, stages %s" % (n, 1e3 * sec, n / sec, st), flush=True)
    run.close()
    del run
    torch.cuda.empty_cache()
