"""The multi-GPU flow rehearsed on ONE device: two processes (gloo for the scalars), each with its own
extractor context on the same GPU, each extracting its shard of a stream of stereo pairs and checking it
against the oracle; then bench.py's own two-rank launch of the sharded EuRoC stream."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, NF, NL, N_FRAMES = 752, 480, 1200, 8, 12


def _frames():
    from orb_slam2_amd import synth
    base = synth.make_batch(N_FRAMES // 2, W + 16, H, 4, 700)
    out = np.zeros((N_FRAMES, H, W), np.uint8)
    out[0::2] = base[:, :, :W]
    out[1::2] = base[:, :, 9:W + 9]     # right image: the scene 9 px further left
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib
    import orb_slam2_amd as oa
    from orb_slam2_amd import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = _frames()
    first, last = shard.frame_shard(N_FRAMES, rank, world)
    assert first % 2 == 0 and last % 2 == 0          # pairs stay on one rank
    ex = oa.ORBextractor(NF, 1.2, NL, 20, 7, device=0, max_width=W, max_height=H, max_batch=last - first, keep_pyramid=False)
    dist.barrier()                                   # both contexts exist: the extractions below overlap on the device
    res = ex.extract_batch(frames[first:last])
    orc = oracle_lib.Oracle(NF, 1.2, NL, 20, 7)
    total = 0
    for i, (kps, desc) in enumerate(res):
        okps, odesc = orc.extract(frames[first + i])
        assert len(kps) == len(okps)
        for f in oa.KP_DTYPE.names:
            assert (kps[f] == okps[f]).all(), (rank, i, f)
        assert (desc == odesc).all(), (rank, i)
        total += len(kps)
    # the stereo stage of the rank's first pair, on the same device context
    u, d = ex.stereo_match(ex, 0.11, 47.9, img_left=0, img_right=1)
    ol, orr = oracle_lib.Oracle(NF, 1.2, NL, 20, 7), oracle_lib.Oracle(NF, 1.2, NL, 20, 7)
    ol.extract(frames[first])
    orr.extract(frames[first + 1])
    ou, od, _, kept = oracle_lib.stereo_match(ol, orr, 0.11, 47.9)
    assert kept > 50 and (u.view(np.uint32) == ou.view(np.uint32)).all() and (d.view(np.uint32) == od.view(np.uint32)).all()
    ex.close()
    s = shard.sum_over_ranks(float(total), dist)
    if rank == 0:
        q.put(s)
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_share_one_device(oracle):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    total = q.get(timeout=10)
    orc = oracle.Oracle(NF, 1.2, NL, 20, 7)
    assert total == float(sum(len(orc.extract(f)[0]) for f in _frames()))


def test_bench_two_rank_stream_rehearsal():
    """bench.py --config c4 under torch.distributed.run with two ranks on the one device (gloo carries the
    scalars): the launch line the driver uses for N > 1, on the strongly scaled EuRoC stream."""
    env = dict(os.environ, ORBX_BENCH_BACKEND="gloo")
    port = 29500 + ((os.getpid() + 977) % 2000)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c4",
                          "--steps", "2", "--warmup", "1", "--no-cpu"], env=env, cwd=ROOT, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    b = json.loads(line)
    assert b["n_gpus"] == 2 and b["scaling"] == "strong" and b["unit"] == "frames/s"
    assert b["config"]["frames_per_step_per_gpu"] == 4096
    assert b["value"] > 1000 and abs(b["value"] - 8192 / (b["ms_per_step"] * 1e-3)) < 1e-6 * b["value"]
    assert 1195 <= b["config"]["keypoints_per_frame"] <= 1230


def test_bench_starts_its_own_ranks():
    """`bench.py --gpus 2` with no launcher and no WORLD_SIZE: it starts the two ranks itself (sharing the one device
    here, gloo for the scalars) and rank 0's line says n_gpus 2; with the RCCL backend and one device it refuses."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--pairs", "16", "--steps", "3", "--warmup",
                          "1", "--no-cpu", "--no-extras"], env=dict(env, ORBX_BENCH_BACKEND="gloo"), cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 alone prints
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["scaling"] == "weak" and b["config"]["frames_per_step_per_gpu"] == 32
    assert abs(b["value"] - 2 * 32 / (b["ms_per_step"] * 1e-3)) < 1e-6 * b["value"]
    import torch
    if torch.cuda.device_count() < 2:           # more ranks than devices over RCCL: an error, never a smaller run
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--no-cpu",
                              "--no-extras"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
        assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
