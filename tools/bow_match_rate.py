#!/usr/bin/env python3
"""Rate of ORBmatcher::SearchByBoW on the benchmark's batch shape: 512 KITTI-size frames (8 different ones, each
shifted by a few pixels from frame to frame) are extracted, their descriptors go through a seeded random
vocabulary of the shape of ORB-SLAM2's (k = 10, L = 6, levelsup = 4: 100 nodes at the matching level), and
every frame is matched as the keyframe against the next one.  Prints milliseconds per batch (run on the GPU
box)."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch  # noqa: E402  (initialises HIP before the library)

import orb_slam2_amd as oa  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402


def random_tree(k, L, seed):
    rng = np.random.default_rng(seed)
    counts = [k ** l for l in range(L + 1)]
    n = sum(counts)
    parent = np.zeros(n, np.int32)
    is_leaf = np.zeros(n, np.uint8)
    first = np.cumsum([0] + counts)
    for l in range(1, L + 1):
        ids = np.arange(first[l], first[l + 1])
        parent[ids] = first[l - 1] + (ids - first[l]) // k
    is_leaf[first[L]:] = 1
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    weight = rng.uniform(0.5, 8.0, n)
    return parent, is_leaf, desc, weight


def main():
    W, H, NF, n = 1241, 376, 2000, 512
    torch.cuda.set_device(0)
    big = synth.make_frame(W + 2 * n, H + 16, 3, 0)
    frames = torch.from_numpy(np.stack([np.ascontiguousarray(big[(i % 8):(i % 8) + H, 2 * i:2 * i + W]) for i in range(n)])).cuda()
    ex = oa.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(W, H)
    d_kps = torch.empty((n, cap, 28), dtype=torch.uint8, device="cuda")
    d_desc = torch.empty((n, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(n, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(frames.data_ptr(), n, W, H, W, W * H, d_kps.data_ptr(), cap, d_desc.data_ptr(), d_n.data_ptr())
    t0 = time.perf_counter()
    voc = oa.ORBVocabulary(10, 6, 0, 0, *random_tree(10, 6, 1))
    print("vocabulary built in %.1f s: %s" % (time.perf_counter() - t0, voc.info()))
    u32 = lambda: torch.empty((n, cap), dtype=torch.int32, device="cuda")  # noqa: E731
    f64 = lambda: torch.empty((n, cap), dtype=torch.float64, device="cuda")  # noqa: E731
    word, node, wt, bid, bval, fnode, fidx = u32(), u32(), f64(), u32(), f64(), u32(), u32()
    bn, fn = torch.empty(n, dtype=torch.int32, device="cuda"), torch.empty(n, dtype=torch.int32, device="cuda")

    def run():
        ex.bow_transform_batch_device(voc, n, d_desc.data_ptr(), cap, d_n.data_ptr(), 4, word.data_ptr(), node.data_ptr(),
                                      wt.data_ptr(), bid.data_ptr(), bval.data_ptr(), bn.data_ptr(), fnode.data_ptr(),
                                      fidx.data_ptr(), fn.data_ptr(), sync=False)
    run()
    ex.synchronize()
    # pair p: frame p as the keyframe (every feature holds a map point), frame p + 1 as the frame
    nxt = torch.roll(torch.arange(n, device="cuda"), -1)
    fnode_f, fidx_f, fn_f = fnode[nxt].contiguous(), fidx[nxt].contiguous(), fn[nxt].contiguous()
    desc_f, kps_f = d_desc[nxt].contiguous(), d_kps[nxt].contiguous()
    valid = torch.ones((n, cap), dtype=torch.uint8, device="cuda")
    match = torch.empty((n, cap), dtype=torch.int32, device="cuda")
    nm = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    def search():
        ex.search_by_bow_batch_device(n, fnode.data_ptr(), fidx.data_ptr(), fn.data_ptr(), d_desc.data_ptr(), d_kps.data_ptr(),
                                      valid.data_ptr(), cap, fnode_f.data_ptr(), fidx_f.data_ptr(), fn_f.data_ptr(),
                                      desc_f.data_ptr(), kps_f.data_ptr(), cap, 0.7, 1, match.data_ptr(), nm.data_ptr(),
                                      sync=False)
    search()
    ex.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        search()
    ex.synchronize()
    ms = (time.perf_counter() - t0) * 100.0
    print("SearchByBoW: %.3f ms per %d pairs (%.0f features, %.0f matches per pair)"
          % (ms, n, float(d_n.float().mean()), float(nm.float().mean())))

    # SearchForTriangulation on the same pairs: no feature holds a map point, no stereo measurements, the views
    # differ by a shift along x (epipolar lines = rows, up to the one-row offset between consecutive views)
    no_mp = torch.zeros((n, cap), dtype=torch.uint8, device="cuda")
    f12 = torch.tensor([[0, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=torch.float32, device="cuda").repeat(n, 1, 1).contiguous()
    epipole = torch.tensor([[-1e5, 188.0]], dtype=torch.float32, device="cuda").repeat(n, 1).contiguous()

    def triangulation():
        ex.search_for_triangulation_batch_device(n, fnode.data_ptr(), fidx.data_ptr(), fn.data_ptr(), d_desc.data_ptr(),
                                                 d_kps.data_ptr(), no_mp.data_ptr(), 0, cap, fnode_f.data_ptr(), fidx_f.data_ptr(),
                                                 fn_f.data_ptr(), desc_f.data_ptr(), kps_f.data_ptr(), no_mp.data_ptr(), 0, cap,
                                                 f12.data_ptr(), epipole.data_ptr(), 0, 1, match.data_ptr(), nm.data_ptr(),
                                                 sync=False)
    triangulation()
    ex.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        triangulation()
    ex.synchronize()
    ms = (time.perf_counter() - t0) * 100.0
    print("SearchForTriangulation: %.3f ms per %d pairs (%.0f matches per pair)" % (ms, n, float(nm.float().mean())))


if __name__ == "__main__":
    main()
