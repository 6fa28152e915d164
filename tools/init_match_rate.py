#!/usr/bin/env python3
"""Rate of ORBmatcher::SearchForInitialization on the benchmark's batch shape: 512 KITTI-size frames (each shifted by
a few pixels from frame to frame) are extracted, undistorted and put on the frame grid, and every frame is matched
against the next one with windowSize 100 and ratio 0.9, vbPrevMatched starting at the frame's own keypoints
(Tracking::MonocularInitialization).  Prints milliseconds per batch (run on the GPU box)."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch  # noqa: E402  (initialises HIP before the library)

import orb_slam2_amd as oa  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402


def main():
    W, H, NF, n = 1241, 376, 2000, 512
    torch.cuda.set_device(0)
    big = synth.make_frame(W + 2 * n, H + 16, 3, 0)
    frames = torch.from_numpy(np.stack([np.ascontiguousarray(big[(i % 8):(i % 8) + H, 2 * i:2 * i + W]) for i in range(n)])).cuda()
    ex = oa.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(W, H)
    d_kps = torch.empty((n, cap, 28), dtype=torch.uint8, device="cuda")
    d_un = torch.empty((n, cap, 28), dtype=torch.uint8, device="cuda")
    d_desc = torch.empty((n, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_cs = torch.empty((n, 64 * 48 + 1), dtype=torch.int32, device="cuda")
    d_ci = torch.empty((n, cap), dtype=torch.int32, device="cuda")
    grid = (0.0, 0.0, float(np.float32(64) / np.float32(W)), float(np.float32(48) / np.float32(H)))
    ex.extract_batch_device(frames.data_ptr(), n, W, H, W, W * H, d_kps.data_ptr(), cap, d_desc.data_ptr(), d_n.data_ptr())
    ex.undistort_grid_batch_device(n, d_kps.data_ptr(), cap, d_n.data_ptr(), np.array([718.856, 718.856, 607.19, 185.2], np.float32),
                                   np.zeros(5, np.float32), grid[0], grid[1], grid[2], grid[3], d_un.data_ptr(), d_cs.data_ptr(),
                                   d_ci.data_ptr(), sync=True)
    nxt = torch.roll(torch.arange(n, device="cuda"), -1)
    un2, cs2, ci2, desc2 = d_un[nxt].contiguous(), d_cs[nxt].contiguous(), d_ci[nxt].contiguous(), d_desc[nxt].contiguous()
    xy = d_un.view(n, cap, 7, 4)[:, :, :2, :].contiguous().view(torch.float32).reshape(n, cap, 2)
    match = torch.empty((n, cap), dtype=torch.int32, device="cuda")
    nm = torch.empty(n, dtype=torch.int32, device="cuda")
    level0 = float((d_un.view(n, cap, 7, 4)[:, :, 5, :].contiguous().view(torch.int32).reshape(n, cap) == 0).float().sum(1).mean())
    torch.cuda.synchronize()

    window = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    check = int(sys.argv[2]) if len(sys.argv) > 2 else 1

    def search(prev):
        ex.search_for_initialization_batch_device(n, d_un.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, prev.data_ptr(),
                                                  un2.data_ptr(), cs2.data_ptr(), ci2.data_ptr(), desc2.data_ptr(), cap, grid[0],
                                                  grid[1], grid[2], grid[3], window, 0.9, check, match.data_ptr(), nm.data_ptr(),
                                                  sync=False)
    prevs = [xy.clone() for _ in range(11)]
    search(prevs[10])
    ex.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        search(prevs[i])
    ex.synchronize()
    ms = (time.perf_counter() - t0) * 100.0
    print("SearchForInitialization (window %d, orientation check %d): %.3f ms per %d pairs (%.0f features, %.0f on level 0, "
          "%.0f matches per pair)" % (window, check, ms, n, float(d_n.float().mean()), level0, float(nm.float().mean())))


if __name__ == "__main__":
    main()
