#!/usr/bin/env python3
"""Rate of the local-map SearchByProjection stage (ORBmatcher::SearchByProjection(F, vpMapPoints, th), the
search of Tracking::SearchLocalPoints) on the benchmark's batch shape: 512 KITTI-size frames are extracted
and bucketed; every frame then receives its own features back as local map points projected a pixel or so
away from where they were seen, on their own or the next pyramid level, th = 1, ratio 0.8.  Prints
milliseconds per batch and the share of points matched (run on the GPU box)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch  # noqa: E402  (initialises HIP before the library)

import orb_slam2_amd as oa  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402


def main():
    W, H, NF, n = 1241, 376, 2000, 512
    K = np.array([718.856, 718.856, 607.1928, 185.2157], np.float32)   # Examples/Stereo/KITTI00-02.yaml
    torch.cuda.set_device(0)
    base = np.stack([synth.make_frame(W, H, 3, i) for i in range(8)])
    frames = torch.from_numpy(np.concatenate([base] * (n // 8))).cuda()
    ex = oa.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(W, H)
    d_kps = torch.empty((n, cap, 28), dtype=torch.uint8, device="cuda")
    d_desc = torch.empty((n, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_un = torch.zeros((n, cap, 28), dtype=torch.uint8, device="cuda")
    d_cs = torch.zeros((n, 64 * 48 + 1), dtype=torch.int32, device="cuda")
    d_ci = torch.zeros((n, cap), dtype=torch.int32, device="cuda")
    ex.extract_batch_device(frames.data_ptr(), n, W, H, W, W * H, d_kps.data_ptr(), cap, d_desc.data_ptr(), d_n.data_ptr())
    inv_w, inv_h = np.float32(64) / np.float32(W), np.float32(48) / np.float32(H)
    ex.undistort_grid_batch_device(n, d_kps.data_ptr(), cap, d_n.data_ptr(), K, np.zeros(5, np.float32), 0.0, 0.0, float(inv_w),
                                   float(inv_h), d_un.data_ptr(), d_cs.data_ptr(), d_ci.data_ptr(), sync=True)
    counts = d_n.cpu().numpy()
    un = d_un.cpu().numpy().reshape(n, cap * 28).view(oa.KP_DTYPE).reshape(n, cap)
    desc = d_desc.cpu().numpy()
    rng = np.random.default_rng(3)
    q = np.zeros((n, cap), oa.LOCAL_QUERY_DTYPE)
    q["u"] = un["x"] + rng.normal(0, 1.0, (n, cap)).astype(np.float32)
    q["v"] = un["y"] + rng.normal(0, 1.0, (n, cap)).astype(np.float32)
    q["level"] = np.clip(un["octave"] + rng.integers(0, 2, (n, cap)), 0, 7)
    q["view_cos"], q["flags"], q["desc"] = 0.99, int(os.environ.get("PROJ_FLAGS", "1")), desc
    th = float(os.environ.get("LOCAL_TH", "1"))
    d_q = torch.from_numpy(q.view(np.uint8).reshape(n, cap * 56)).cuda()
    d_match = torch.empty((n, cap), dtype=torch.int32, device="cuda")
    d_qm = torch.empty((n, cap), dtype=torch.int32, device="cuda")
    d_nm = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    def run():
        ex.search_local_map_batch_device(n, d_un.data_ptr(), cap, d_cs.data_ptr(), d_ci.data_ptr(), d_desc.data_ptr(), 0, 0, 0.0,
                                         0.0, float(inv_w), float(inv_h), th, 0.8, d_q.data_ptr(), cap, d_n.data_ptr(),
                                         d_match.data_ptr(), d_qm.data_ptr(), d_nm.data_ptr(), sync=False)
    run()
    ex.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run()
    ex.synchronize()
    ms = (time.perf_counter() - t0) * 100.0
    print("local-map SearchByProjection, th %.0f: %.3f ms per %d frames (%.0f map points per frame, %.1f %% matched)"
          % (th, ms, n, counts.mean(), 100.0 * float(d_nm.float().sum()) / counts.sum()))


if __name__ == "__main__":
    main()
