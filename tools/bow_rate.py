#!/usr/bin/env python3
"""Rate of the bag-of-words stage on the benchmark's batch shape: 512 KITTI-size frames are extracted,
then their descriptors go through a seeded random vocabulary of the shape of ORB-SLAM2's (k = 10,
L = 6: 1.1 M nodes, 1 M words), levelsup = 4.  Prints milliseconds per batch (run on the GPU box)."""
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
    base = np.stack([synth.make_frame(W, H, 3, i) for i in range(8)])
    frames = torch.from_numpy(np.concatenate([base] * (n // 8))).cuda()
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
    t0 = time.perf_counter()
    for _ in range(10):
        run()
    ex.synchronize()
    ms = (time.perf_counter() - t0) * 100.0
    print("bag-of-words transform: %.3f ms per %d frames (%.0f features, %.1f words per frame)"
          % (ms, n, float(d_n.float().mean()), float(bn.float().mean())))


if __name__ == "__main__":
    main()
