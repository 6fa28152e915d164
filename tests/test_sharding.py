"""The multi-GPU path on CPU: two gloo ranks shard a stream of stereo pairs, extract their share
(with the oracle standing in for the device), and only scalars cross ranks."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from orb_slam2_amd import shard, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pair_shard_partitions():
    for n in (1, 2, 7, 256, 4096):
        for world in (1, 2, 3, 4, 8):
            spans = [shard.pair_shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard.frame_shard(16, 1, 2) == (8, 16)


def _worker(rank, world, port, n_frames, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, last = shard.frame_shard(n_frames, rank, world)
    orc = oracle_lib.Oracle(300, 1.2, 4, 20, 7)
    total = 0
    for i in range(first, last):
        kps, _ = orc.extract(synth.make_frame(200, 160, 9, i))
        total += len(kps)
    dist.barrier()
    t = shard.max_over_ranks(float(rank + 1), dist)
    s = shard.sum_over_ranks(float(total), dist)
    if rank == 0:
        out.put((t, s, first, last))
    dist.destroy_process_group()


def test_two_rank_gloo_job(oracle):
    n_frames, world = 8, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    t, s, first, last = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert t == 2.0 and (first, last) == (0, 4)
    orc = oracle.Oracle(300, 1.2, 4, 20, 7)
    expect = sum(len(orc.extract(synth.make_frame(200, 160, 9, i))[0]) for i in range(n_frames))
    assert s == float(expect) and expect > 0


def test_library_frame_shard_equals_the_python_cut():
    """orbx_host_frame_shard (what the in-process worker pool cuts a stream with) against shard.frame_shard: a
    partition into contiguous runs, pairs never split, sizes within one pair of each other."""
    import orb_slam2_amd as oa
    for n in (0, 1, 2, 3, 7, 8, 64, 511, 512, 8192):
        for world in (1, 2, 3, 4, 8, 16):
            spans = [oa.host_frame_shard(n, r, world) for r in range(world)]
            assert spans == [shard.frame_shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            for a, b in spans:
                assert (a % 2 == 0 or a == n) and (b % 2 == 0 or b == n)
            pairs = [(b - a + 1) // 2 for a, b in spans]       # an odd stream ends in half a pair
            assert max(pairs) - min(pairs) <= 1
    L = oa.load_library()
    import ctypes as C
    a, b = C.c_int(), C.c_int()
    assert L.orbx_host_frame_shard(8, 2, 2, C.byref(a), C.byref(b)) == -1
    assert L.orbx_host_frame_shard(8, 0, 0, C.byref(a), C.byref(b)) == -1
    assert L.orbx_host_frame_shard(-1, 0, 1, C.byref(a), C.byref(b)) == -1
