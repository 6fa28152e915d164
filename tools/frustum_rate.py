#!/usr/bin/env python3
"""Rate of the map-point projections (Frame::isInFrustum with MapPoint::PredictScale, and the tests in front of
ORBmatcher::Fuse's search) on the benchmark's batch shape: 512 frames with 2000 map points each, scattered in front
of the camera.  68 bytes are read and 56 written per point.  Prints milliseconds per batch and the rate against the
bytes moved (run on the GPU box)."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch  # noqa: E402  (initialises HIP before the library)

import orb_slam2_amd as oa  # noqa: E402


def main():
    n, q_cap = 512, 2000
    torch.cuda.set_device(0)
    ex = oa.ORBextractor(2000, 1.2, 8, 20, 7, max_width=1241, max_height=376, max_batch=1, keep_pyramid=False)
    rng = np.random.default_rng(2)
    pts = np.zeros((n, q_cap), oa.FRUSTUM_POINT_DTYPE)
    z = rng.uniform(1, 30, (n, q_cap))
    pts["x"], pts["y"], pts["z"] = rng.uniform(-0.9, 0.9, (n, q_cap)) * z, rng.uniform(-0.3, 0.3, (n, q_cap)) * z, z
    pts["nz"] = 1.0
    dist = np.sqrt(pts["x"].astype(np.float64) ** 2 + pts["y"].astype(np.float64) ** 2 + z ** 2)
    pts["max_dist"] = dist * 1.2 ** rng.uniform(0, 7, (n, q_cap))
    pts["min_dist"] = pts["max_dist"] / 1.2 ** 7
    pts["flags"] = 1
    poses = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (n, 1))
    d_pts = torch.from_numpy(pts.view(np.uint8).reshape(n, q_cap * 68)).cuda()
    d_pose, d_ow = torch.from_numpy(poses).cuda(), torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    d_nq = torch.full((n,), q_cap, dtype=torch.int32, device="cuda")
    d_q = torch.empty((n, q_cap, 56), dtype=torch.uint8, device="cuda")
    d_cnt = torch.empty(n, dtype=torch.int32, device="cuda")
    p = oa.FrustumParams()
    p.fx, p.fy, p.cx, p.cy, p.mbf = 718.856, 718.856, 607.1928, 185.2157, 386.1448
    p.min_x, p.max_x, p.min_y, p.max_y = 0.0, 1241.0, 0.0, 376.0
    p.view_cos_limit, p.th = 0.5, 3.0
    p.log_scale_factor = float(np.float32(np.log(np.float64(np.float32(1.2)))))
    for name, fn in (("isInFrustum", ex.is_in_frustum_batch_device), ("Fuse projection", ex.fuse_project_batch_device)):
        def run():
            fn(n, p, d_pose.data_ptr(), d_ow.data_ptr(), d_pts.data_ptr(), q_cap, d_nq.data_ptr(), d_q.data_ptr(),
               d_cnt.data_ptr(), sync=False)
        run()
        ex.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            run()
        ex.synchronize()
        ms = (time.perf_counter() - t0) * 50.0
        print("%s: %.4f ms per %d frames x %d points (%.0f %% in view), %.2f TB/s of the 124 bytes per point"
              % (name, ms, n, q_cap, 100.0 * float(d_cnt.float().sum()) / (n * q_cap), n * q_cap * 124 / ms / 1e9))


if __name__ == "__main__":
    main()
