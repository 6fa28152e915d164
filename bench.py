#!/usr/bin/env python3
"""Benchmark of the ORB extractor hot path on MI355X.

One step = one pass of the whole path (pyramid, FAST cells, quadtree retention, blur,
orientation + descriptors) over one batch of KITTI-shape frames: 256 stereo pairs = 512 frames of
1241x376, 2000 features, 8 levels, scale 1.2, FAST 20/7, with the frames already resident in
HBM and the results left in HBM.  N > 1: one process per GPU, every rank runs its own batch
(frames are independent: no collective on the data path); value = frames of all ranks / max
time over ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H, NFEAT, NLEVELS, SCALE, INI_TH, MIN_TH = 1241, 376, 2000, 8, 1.2, 20, 7
PAIRS = 256
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
# HBM bytes per launch from the committed rocprofv3 counter passes of this same command
# (tools/traffic_pass.sh + tools/traffic_summary.py); counters cannot be read from inside the process.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r01_traffic.json")
STAGE_KERNEL = {"pyramid": "k_resize_rows", "fast": "k_fast_tiles", "quadtree": "k_octree", "blur": "k_blur7_all",
                "orient_describe": "k_orient_describe"}


def measured_traffic(stage, frames_per_step, launches):
    """PMC-measured HBM bytes per launch of the stage's kernel (its bytes per 512-frame step over its launches
    per step), scaled to this run's batch size."""
    try:
        with open(TRAFFIC_FILE) as f:
            t = json.load(f)
        return t[STAGE_KERNEL[stage]]["traffic_bytes"] / launches * frames_per_step / 512.0
    except (OSError, KeyError, ValueError):
        return None


VALU_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_valu.json")
N_SIMD, VALU_CYCLES, CLOCK_HZ = 256 * 4, 4.0, 2.4e9  # a SIMD issues one wave64 vector instruction per 4 cycles


def valu_view(stage, frames_per_step, serial_ms):
    """The stage's kernel against the vector-issue ceiling: its counted wave instructions per step
    (SQ_INSTS_VALU, profiles/) at one instruction per SIMD per 4 cycles, beside its measured time alone."""
    try:
        with open(VALU_FILE) as f:
            t = json.load(f)
        instr = t[STAGE_KERNEL[stage]]["SQ_INSTS_VALU"] * frames_per_step / 512.0
    except (OSError, KeyError, ValueError):
        return None
    floor_ms = instr * VALU_CYCLES / N_SIMD / CLOCK_HZ * 1e3
    return {"wave_instructions_per_step": instr, "issue_floor_ms": floor_ms, "serialised_ms": serial_ms,
            "frac_of_issue_ceiling": floor_ms / serial_ms if serial_ms > 0 else None}


def level_pixels(oa):
    px = []
    for l in range(NLEVELS):
        g = oa.host_level_geometry(NFEAT, SCALE, NLEVELS, W, H, l)
        px.append(g.width * g.height)
    return px


def make_frames_gpu(torch, n, rank, stride):
    """n synthetic frames in HBM as n/2 stereo pairs (2k, 2k+1).  Left frames: n/8 generated scenes x
    {vertical flip} x {inversion}; the right frame of a pair is its left frame seen 4..19 px further
    left (a constant-disparity view, columns beyond the edge repeated), so that the stereo stage after
    the extractor has real correspondences to find."""
    from orb_slam2_amd import synth
    nbase = (n + 7) // 8
    base = synth.make_batch(nbase, W, H, config_id=3, first_idx=rank * nbase)
    tb = torch.from_numpy(base).cuda()
    lefts = []
    for inv in (False, True):
        for fv in (False, True):
            t = tb
            if fv:
                t = t.flip(1)
            if inv:
                t = 255 - t
            lefts.append(t)
    left = torch.stack(lefts, 1).reshape(-1, H, W)[:(n + 1) // 2]
    right = torch.empty_like(left)
    for k in range(left.shape[0]):
        d = 4 + (k % 16)
        right[k, :, :W - d] = left[k, :, d:]
        right[k, :, W - d:] = left[k, :, W - 1:W]
    allf = torch.stack([left, right], 1).reshape(-1, H, W)[:n]
    frames = torch.zeros((n, H, stride), dtype=torch.uint8, device="cuda")
    frames[:, :, :W] = allf
    return frames, base


def cpu_baseline(base_frames, budget_s=12.0):
    """The CPU oracle (a restatement of the reference's ORBextractor) on one host core."""
    import oracle_lib
    orc = oracle_lib.Oracle(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH)
    orc.extract(base_frames[0])  # warm-up
    t0 = time.perf_counter()
    done = 0
    while (time.perf_counter() - t0) < budget_s * 0.8:   # about 10 s: the generated frames, cycled
        orc.extract(base_frames[done % len(base_frames)])
        done += 1
    dt = time.perf_counter() - t0
    out = {"value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d extractions over the %d generated 1241x376 frames (cycled), one at a time, single thread, "
                     "oracle/liborb_oracle.so (g++ -O2)" % (done, len(base_frames)),
           "ms_per_frame": 1000.0 * dt / done}
    # the same oracle, frame-parallel over the host cores this job may use (one instance per thread;
    # the C call releases the interpreter lock)
    import threading
    ncores = max(1, min(16, len(os.sched_getaffinity(0))))
    counts = [0] * ncores

    def worker(i):
        o = oracle_lib.Oracle(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH)
        k = i
        while time.perf_counter() - t1 < budget_s * 0.5:
            o.extract(base_frames[k % len(base_frames)])
            k += ncores
            counts[i] += 1

    t1 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(ncores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt_all = time.perf_counter() - t1
    out["all_cores"] = {"value": sum(counts) / dt_all, "unit": "frames/s", "cores": ncores,
                        "sample": "%d frames over %d threads" % (sum(counts), ncores)}
    return out


def projection_stage(torch, oa, ex, n, cap, d_kps, d_desc, d_n, steps):
    """Times orbx_search_by_projection_batch_device on the batch the extractor just produced."""
    fx, fy, cx, cy = 718.856, 718.856, 607.1928, 185.2157  # Examples/Stereo/KITTI00-02.yaml (rectified: no distortion)
    d_un = torch.zeros((n, cap, 28), dtype=torch.uint8, device="cuda")
    d_cs = torch.zeros((n, 64 * 48 + 1), dtype=torch.int32, device="cuda")
    d_ci = torch.zeros((n, cap), dtype=torch.int32, device="cuda")
    inv_w, inv_h = float(np.float32(64) / np.float32(W)), float(np.float32(48) / np.float32(H))
    ex.undistort_grid_batch_device(n, d_kps.data_ptr(), cap, d_n.data_ptr(), np.array([fx, fy, cx, cy], np.float32),
                                   np.zeros(5, np.float32), 0.0, 0.0, inv_w, inv_h, d_un.data_ptr(), d_cs.data_ptr(),
                                   d_ci.data_ptr(), sync=True)
    kp = d_un.view(torch.float32)                      # x, y, size, angle, response, octave, class_id
    g = torch.Generator(device="cuda").manual_seed(7)
    z = 4.0 + 36.0 * torch.rand((n, cap), device="cuda", generator=g)
    q = torch.zeros((n, cap, 14), dtype=torch.float32, device="cuda")   # 56-byte orbx_map_query records
    q[..., 0] = (kp[..., 0] - cx) * z / fx
    q[..., 1] = (kp[..., 1] - cy) * z / fy
    q[..., 2] = z
    q[..., 4] = kp[..., 3]
    qi = q.view(torch.int32)
    qi[..., 3] = d_un.view(torch.int32)[..., 5]
    qi[..., 5] = 1                                      # every map point has observations
    q.view(torch.uint8)[..., 24:56] = d_desc
    poses = torch.zeros((n, 24), dtype=torch.float32, device="cuda")
    poses[:, [0, 5, 10, 12, 17, 22]] = 1.0
    poses[:, 3] = 0.02 * torch.randn(n, device="cuda", generator=g)
    poses[:, 11] = -0.3
    P = np.zeros((), oa.PROJECTION_PARAMS_DTYPE)
    P["fx"], P["fy"], P["cx"], P["cy"], P["mbf"], P["mb"] = fx, fy, cx, cy, 386.1448, 386.1448 / fx
    P["min_x"], P["min_y"], P["max_x"], P["max_y"], P["inv_w"], P["inv_h"] = 0, 0, W, H, inv_w, inv_h
    P["th"], P["mono"], P["check_orientation"] = 15.0, 0, 1
    d_match = torch.empty((n, cap), dtype=torch.int32, device="cuda")
    d_qm = torch.empty((n, cap), dtype=torch.int32, device="cuda")
    d_nm = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    def run():
        ex.search_by_projection_batch_device(n, d_un.data_ptr(), cap, d_cs.data_ptr(), d_ci.data_ptr(), d_desc.data_ptr(), 0, 0,
                                             P, poses.data_ptr(), q.data_ptr(), cap, d_n.data_ptr(), d_match.data_ptr(),
                                             d_qm.data_ptr(), d_nm.data_ptr(), sync=False)
    run()
    ex.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    ex.synchronize()
    ms = 1000.0 * (time.perf_counter() - t0) / steps
    return {"name": "ORBmatcher::SearchByProjection(current, last) for every frame of the batch, th 15, rotation check "
                    "(not part of value)",
            "ms_per_step": ms, "frames_per_s": n / (ms * 1e-3),
            "matched_fraction": float(d_nm.sum().item()) / max(1.0, float(d_n.sum().item()))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=PAIRS)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import orb_slam2_amd as oa
    from orb_slam2_amd import shard

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # ORBX_BENCH_BACKEND=gloo rehearses the multi-process flow on a box with fewer GPUs than ranks
    # (ranks then share devices); the measured configuration is one rank per GPU over RCCL ("nccl").
    backend = os.environ.get("ORBX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    red_dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=red_dev)
        else:
            dist.init_process_group(backend)

    n = 2 * args.pairs
    stride = (W + 63) // 64 * 64
    frames, base = make_frames_gpu(torch, n, rank, stride)
    ex = oa.ORBextractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, device=local_rank, max_width=W, max_height=H,
                         max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(W, H)
    d_kps = torch.empty((n, cap, 28), dtype=torch.uint8, device="cuda")
    d_desc = torch.empty((n, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(n, dtype=torch.int32, device="cuda")

    def step():
        ex.extract_batch_device(frames.data_ptr(), n, W, H, stride, H * stride, d_kps.data_ptr(), cap,
                                d_desc.data_ptr(), d_n.data_ptr(), sync=False)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    ex.synchronize()
    ex.set_profiling(True)
    stage_sum = {}
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()          # steps are enqueued back to back: a batch's descriptors overlap the next batch's FAST
    ex.synchronize()    # the extractor's own streams (torch's current stream never sees them)
    # HIP events recorded on the extractor's streams around each stage's launches; the events of the last
    # step of the timed region are the ones left to read
    stage_sum = {k: v * args.steps for k, v in ex.stage_ms().items()}
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = shard.max_over_ranks(time.perf_counter() - t0, dist, red_dev)  # the job is as slow as its slowest rank

    # Outside the timed region: the same steps with the stages serialised on one stream, so that each
    # kernel's duration is its own (in the timed region the pyramid and the blur run beside FAST and the
    # quadtree on a second stream and stretch each other).
    ex.set_debug_flag(oa.FLAG_NO_OVERLAP, 1)
    serial_sum = {}
    step()
    ex.synchronize()
    for _ in range(min(args.steps, 5)):
        step()
        ex.synchronize()
        for k, v in ex.stage_ms().items():
            serial_sum[k] = serial_sum.get(k, 0.0) + v / min(args.steps, 5)
    ex.set_debug_flag(oa.FLAG_NO_OVERLAP, 0)

    # the stage after the extractor (stereo association of the pairs of the batch), timed on its own:
    # it is not part of `value`
    d_u = torch.empty((args.pairs, cap), dtype=torch.float32, device="cuda")
    d_dep = torch.empty((args.pairs, cap), dtype=torch.float32, device="cuda")
    mb, mbf = 0.5371657, 386.1448  # KITTI 00-02
    torch.cuda.synchronize()

    def stereo():
        ex.stereo_match_batch_device(args.pairs, d_kps.data_ptr(), cap, d_desc.data_ptr(), d_n.data_ptr(), mb, mbf,
                                     d_u.data_ptr(), d_dep.data_ptr(), sync=False)

    stereo()
    ex.synchronize()
    ts = time.perf_counter()
    for _ in range(args.steps):
        stereo()
    ex.synchronize()
    stereo_ms = 1000.0 * (time.perf_counter() - ts) / args.steps

    # a descriptor consumer further down the tracking thread, timed on its own (not part of `value`):
    # ORBmatcher::SearchByProjection between consecutive frames, every frame given its own features back
    # as the last frame's map points, seen from a pose 30 cm further on
    proj = None
    if rank == 0:
        try:
            proj = projection_stage(torch, oa, ex, n, cap, d_kps, d_desc, d_n, args.steps)
        except Exception as e:  # the headline measurement does not depend on this stage
            proj = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        counts = d_n.cpu().numpy()
        kavg = float(counts.mean())
        px = level_pixels(oa)
        P = sum(px)
        b_frame = 5 * P - px[0] - px[-1] + 1321 * kavg
        fps = world * n * args.steps / dt
        stage_ms = {k: v / args.steps for k, v in stage_sum.items()}
        # compulsory bytes per frame of each stage (read once, write once)
        stage_bytes = {
            "pyramid": (P - px[-1]) + (P - px[0]),
            "fast": P,
            "quadtree": 0,  # candidate lists: outside the algorithmic byte count
            "blur": 2 * P,
            "orient_describe": 1321 * kavg,
        }
        dom = max((k for k in stage_ms if stage_bytes[k] > 0), key=lambda k: stage_ms[k])
        launches = {"pyramid": NLEVELS - 1, "fast": 2, "quadtree": 1, "blur": 1, "orient_describe": 1}
        # per launch: algorithmic bytes of the stage / its launches, over the average launch duration
        dom_bytes = stage_bytes[dom] * n / launches[dom]
        achieved = dom_bytes / (stage_ms[dom] / launches[dom] * 1e-3) / 1e9
        out = {
            "metric": "ORB frames/sec (1241x376, 2000 kp, 8 lvl)",
            "value": fps,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "KITTI stereo 1241x376 left+right, 2000 features/image, 8 levels, scale 1.2, "
                                   "FAST 20/7, %d pairs (%d frames) per step per GPU, frames resident in HBM"
                                   % (args.pairs, n),
                       "frames_per_step_per_gpu": n, "keypoints_per_frame": kavg, "parallelism": "frames sharded "
                       "across GPUs, no collective"},
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "launches_per_step": launches[dom],
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(dom, n, launches[dom]),
                "algorithmic_bytes_per_launch": dom_bytes,
                "kernel_ms_per_launch": stage_ms[dom] / launches[dom],
                "note": "timed region: FAST runs as 2 launches (level 0, then levels 1..7) while the pyramid and the blur "
                        "run beside it on a second stream; serialised_* are the same kernels alone on the GPU "
                        "(measured after the timed region)",
                "serialised_ms_per_launch": serial_sum[dom] / (1 if dom == "fast" else launches[dom]),
                "serialised_frac": stage_bytes[dom] * n / (serial_sum[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "valu_issue": valu_view(dom, n, serial_sum[dom]),
            },
            "path_roofline": {
                "algorithmic_bytes_per_frame": b_frame,
                "achieved_GBs_per_gpu": b_frame * fps / world / 1e9,
                "frac_of_hbm_peak": b_frame * fps / world / 1e9 / HBM_PEAK_GBS,
            },
            "stage_ms": stage_ms,
            "stage_ms_serialised": serial_sum,
            "next_stage": {"name": "stereo association of the batch's pairs (not part of value)",
                           "ms_per_step": stereo_ms, "pairs_per_s": args.pairs / (stereo_ms * 1e-3),
                           "matched_fraction": float((d_u >= 0).float().sum().item() / max(1.0, float(counts[0::2].sum())))},
            "next_stage_matching": proj,
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(base)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    ex.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
