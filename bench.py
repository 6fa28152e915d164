#!/usr/bin/env python3
"""Benchmark of the ORB extractor hot path on MI355X.

One step = one pass of the whole path (pyramid, FAST cells, quadtree retention, blur,
orientation + descriptors) over one batch of KITTI-shape frames: 256 stereo pairs = 512 frames of
1241x376, 2000 features, 8 levels, scale 1.2, FAST 20/7, with the frames already resident in
HBM and the results left in HBM.  N > 1: one process per GPU, every rank runs its own batch
(frames are independent: no collective on the data path); value = frames of all ranks / max
time over ranks.

`--config c4` measures the other multi-GPU configuration instead: a stream of 8192 EuRoC-shape
frames (752x480, 1200 features) cut by orb_slam2_amd.shard.frame_shard over the ranks (strong
scaling: the stream is fixed, a step is one pass over the whole stream).

Prints ONE JSON line on rank 0.  Beside the headline fields it carries, each measured in this same
run: per-block statistics of the step time, the stage times of the kernels alone on the GPU, the
PCIe-inclusive rate of the host-memory batch entry point, the single-frame latency and the batched
rate of the 640x480 configuration, the 4K configuration, the EuRoC stream on this GPU, two
consumers of the batch (stereo association, SearchByProjection), and the CPU oracle on the host.
"""
import argparse
import ctypes as C
import glob
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCALE, INI_TH, MIN_TH = 1.2, 20, 7
C3 = dict(name="c3", w=1241, h=376, nfeat=2000, nlevels=8, config_id=3)   # KITTI (Examples/Stereo/KITTI00-02.yaml)
C2 = dict(name="c2", w=640, h=480, nfeat=1000, nlevels=8, config_id=1)    # TUM (Examples/Monocular/TUM1.yaml)
C4 = dict(name="c4", w=752, h=480, nfeat=1200, nlevels=8, config_id=4)    # EuRoC (Examples/Stereo/EuRoC.yaml)
C5 = dict(name="c5", w=3840, h=2160, nfeat=8000, nlevels=12, config_id=5)
PAIRS = 256
PROFILED_FRAMES = {"c2": 256, "c3": 512, "c4": 512, "c5": 128}   # frames per step of the commands the counter files were collected on
PROFILED_FRAMES_NOW = [512]
C4_STREAM = 8192
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
STAGE_KERNEL = {"pyramid": "k_resize_rows", "fast": "k_fast_tiles", "quadtree": "k_octree", "blur": "k_blur7_all",
                "orient_describe": "k_orient_describe"}
N_SIMD, VALU_CYCLES, CLOCK_HZ = 256 * 4, 4.0, 2.4e9  # a SIMD issues one wave64 vector instruction per 4 cycles


COUNTER_PREFIX = ""   # "c5_" while the 4K configuration is the measured step: its counter files are profiles/rNN_c5_<kind>.json


def counter_file(kind):
    kind = COUNTER_PREFIX + kind
    return _counter_file(kind)


def _counter_file(kind):
    """The newest profiles/rNN_<kind>.json (counter passes of this same command, collected with
    tools/traffic_pass.sh / tools/pmc_multi.sh: counters cannot be read from inside the process).  A file
    made from other kernel sources than the ones loaded now is refused: its numbers describe other code."""
    import orb_slam2_amd as oa
    best = None
    for p in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s.json" % kind)):
        r = int(re.search(r"r(\d\d)_", os.path.basename(p)).group(1))
        if best is None or r > best[0]:
            best = (r, p)
    if best is None:
        return None, "no profiles/rNN_%s.json" % kind
    try:
        with open(best[1]) as f:
            t = json.load(f)
    except (OSError, ValueError) as e:
        return None, "%s: %s" % (os.path.basename(best[1]), e)
    if t.get("_source_sha256") != oa.source_hash():
        return None, "%s was collected on other kernel sources (hash differs): not reported" % os.path.basename(best[1])
    return t, os.path.basename(best[1])


def measured_traffic(stage, frames_per_step, launches):
    """PMC-measured HBM bytes per launch of the stage's kernel (its bytes per 512-frame step over its launches
    per step), scaled to this run's batch size; None when no counter file matches the loaded kernels."""
    t, src = counter_file("traffic")
    if t is None:
        return None, src
    try:
        return t[STAGE_KERNEL[stage]]["traffic_bytes"] / launches * frames_per_step / PROFILED_FRAMES_NOW[0], src
    except KeyError:
        return None, src


def valu_view(stage, frames_per_step, serial_ms):
    """The stage's kernel against the vector-issue ceiling: its counted wave instructions per step
    (SQ_INSTS_VALU) at one instruction per SIMD per 4 cycles, beside its measured time alone."""
    t, src = counter_file("valu")
    if t is None:
        return {"note": src}
    try:
        instr = t[STAGE_KERNEL[stage]]["SQ_INSTS_VALU"] * frames_per_step / PROFILED_FRAMES_NOW[0]
    except KeyError:
        return {"note": src}
    floor_ms = instr * VALU_CYCLES / N_SIMD / CLOCK_HZ * 1e3
    return {"wave_instructions_per_step": instr, "issue_floor_ms": floor_ms, "serialised_ms": serial_ms,
            "frac_of_issue_ceiling": floor_ms / serial_ms if serial_ms > 0 else None, "source": src}


def path_valu_view(frames_per_step, ms_per_step, frames_profiled):
    """The whole step against the vector-issue ceiling, which is what bounds it: the five kernels' counted wave
    instructions per step (SQ_INSTS_VALU) at one instruction per SIMD per 4 cycles, beside the measured step; and
    per kernel the share of the LDS array's cycles lost to bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)."""
    t, src = counter_file("valu")
    if t is None:
        return {"note": src}
    per, lds, total = {}, {}, 0.0
    for stage, kern in STAGE_KERNEL.items():
        # a stage may run as more than one kernel (the pyramid of a single frame): every kernel of the file counts
        names = [k for k in t if not k.startswith("_") and (k == kern or (stage == "pyramid" and k.startswith("k_resize")) or
                                                            (stage == "quadtree" and k.startswith("k_octree")))]
        v = sum(t[k].get("SQ_INSTS_VALU", 0.0) for k in names) * frames_per_step / frames_profiled
        per[stage] = v
        total += v
        act = sum(t[k].get("SQ_LDS_IDX_ACTIVE", 0.0) for k in names)
        lds[stage] = (sum(t[k].get("SQ_LDS_BANK_CONFLICT", 0.0) for k in names) / act) if act > 0 else 0.0
    floor_ms = total * VALU_CYCLES / N_SIMD / CLOCK_HZ * 1e3
    return {"wave_instructions_per_step": total, "per_kernel": per, "issue_floor_ms": floor_ms, "ms_per_step": ms_per_step,
            "step_over_floor": ms_per_step / floor_ms if floor_ms > 0 else None, "lds_conflict_frac": lds, "source": src,
            "note": "the step is bound by the sum of its kernels' vector instructions: floor = instructions x 4 cycles / 1024 "
                    "SIMDs / 2.4 GHz; lds_conflict_frac = bank-conflict cycles / LDS array cycles per kernel"}


def level_pixels(oa, cfg):
    px = []
    for l in range(cfg["nlevels"]):
        g = oa.host_level_geometry(cfg["nfeat"], SCALE, cfg["nlevels"], cfg["w"], cfg["h"], l)
        px.append(g.width * g.height)
    return px


def frame_bytes(oa, cfg, kavg):
    """Algorithmic bytes per frame: 5P - p_0 - p_last + 1321 K (each stage reads its compulsory input once and
    writes its output once)."""
    px = level_pixels(oa, cfg)
    P = sum(px)
    return 5 * P - px[0] - px[-1] + 1321 * kavg, px


def make_frames_gpu(torch, cfg, n, first_idx, stride, n_base=None):
    """n synthetic frames of cfg's shape in HBM as n/2 stereo pairs (2k, 2k+1).  Left frames: generated
    scenes x {vertical flip} x {inversion}; the right frame of a pair is its left frame seen 4..19 px
    further left (a constant-disparity view, columns beyond the edge repeated), so that the stereo stage
    after the extractor has real correspondences to find."""
    from orb_slam2_amd import synth
    W, H = cfg["w"], cfg["h"]
    nbase = n_base or (n + 7) // 8
    base = synth.make_batch(nbase, W, H, config_id=cfg["config_id"], first_idx=first_idx)
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
    left = torch.stack(lefts, 1).reshape(-1, H, W)
    reps = ((n + 1) // 2 + left.shape[0] - 1) // left.shape[0]
    left = left.repeat(reps, 1, 1)[:(n + 1) // 2]
    right = torch.empty_like(left)
    for k in range(left.shape[0]):
        d = 4 + (k % 16)
        right[k, :, :W - d] = left[k, :, d:]
        right[k, :, W - d:] = left[k, :, W - 1:W]
    allf = torch.stack([left, right], 1).reshape(-1, H, W)[:n]
    frames = torch.zeros((n, H, stride), dtype=torch.uint8, device="cuda")
    frames[:, :, :W] = allf
    torch.cuda.synchronize()
    return frames, base


class DeviceRun:
    """One extractor context with a resident batch and result arrays (torch only allocates)."""

    def __init__(self, torch, oa, cfg, n, device, first_idx=0, n_base=None):
        self.cfg, self.n = cfg, n
        W, H = cfg["w"], cfg["h"]
        self.stride = (W + 63) // 64 * 64
        self.frames, self.base = make_frames_gpu(torch, cfg, n, first_idx, self.stride, n_base)
        self.ex = oa.ORBextractor(cfg["nfeat"], SCALE, cfg["nlevels"], INI_TH, MIN_TH, device=device, max_width=W,
                                  max_height=H, max_batch=n, keep_pyramid=False)
        self.cap = self.ex.max_keypoints(W, H)
        self.d_kps = torch.empty((n, self.cap, 28), dtype=torch.uint8, device="cuda")
        self.d_desc = torch.empty((n, self.cap, 32), dtype=torch.uint8, device="cuda")
        self.d_n = torch.zeros(n, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()   # torch's fills run on its own stream: done before the extractor's streams write

    def step(self, first=0, count=None):
        count = self.n - first if count is None else count
        H = self.cfg["h"]
        self.ex.extract_batch_device(self.frames.data_ptr() + first * H * self.stride, count, self.cfg["w"], H, self.stride,
                                     H * self.stride, self.d_kps.data_ptr() + first * self.cap * 28, self.cap,
                                     self.d_desc.data_ptr() + first * self.cap * 32, self.d_n.data_ptr() + 4 * first,
                                     sync=False)

    def close(self):
        self.ex.close()


def timed_steps(run, steps, warmup, fn=None):
    fn = fn or run.step
    for _ in range(warmup):
        fn()
    run.ex.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    run.ex.synchronize()
    return (time.perf_counter() - t0) / steps


def build_native_oracle():
    """The oracle once more with -march=native, built on the machine that runs it (the committed build is
    plain -O2: an -march=native object made elsewhere may not run here).  Returns the path or None."""
    try:
        out = os.path.join(tempfile.mkdtemp(prefix="orbx_oracle_"), "liborb_oracle_native.so")
        subprocess.check_call(["g++", "-O2", "-march=native", "-std=c++14", "-fPIC", "-ffp-contract=off", "-shared", "-o", out,
                               os.path.join(ROOT, "oracle", "orb_oracle.cpp")], timeout=180,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return out
    except Exception:
        return None


def cpu_baseline(base_frames, cfg, repeats=5, repeat_s=2.0):
    """The CPU oracle (a restatement of the reference's ORBextractor) on the host: one thread, one frame at a
    time (how the reference runs), median of `repeats` timed runs of about `repeat_s` seconds each, for the
    committed -O2 build and for an -march=native build made here; then frame-parallel over the host cores."""
    import oracle_lib

    def one_core(lib_path, label):
        orc = oracle_lib.Oracle(cfg["nfeat"], SCALE, cfg["nlevels"], INI_TH, MIN_TH, lib_path=lib_path)
        orc.extract(base_frames[0])  # warm-up
        rates, done_all = [], 0
        for _ in range(repeats):
            t0, done = time.perf_counter(), 0
            while time.perf_counter() - t0 < repeat_s:
                orc.extract(base_frames[(done_all + done) % len(base_frames)])
                done += 1
            rates.append(done / (time.perf_counter() - t0))
            done_all += done
        return {"value": float(np.median(rates)), "min": min(rates), "max": max(rates), "frames": done_all, "build": label}

    o2 = one_core(None, "g++ -O2 (oracle/liborb_oracle.so)")
    out = {"value": o2["value"], "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "median of %d runs of about %.0f s each (%d extractions in all) over the %d generated %dx%d frames, cycled, "
                     "one at a time, single thread, %s" % (repeats, repeat_s, o2["frames"], len(base_frames), cfg["w"], cfg["h"],
                                                           o2["build"]),
           "ms_per_frame": 1000.0 / o2["value"], "spread": {"min": o2["min"], "max": o2["max"]}}
    native = build_native_oracle()
    if native:
        try:
            nv = one_core(native, "g++ -O2 -march=native, built on this host")
            out["march_native"] = {"value": nv["value"], "unit": "frames/s", "cores": 1,
                                   "spread": {"min": nv["min"], "max": nv["max"]}, "sample": "same sampling, %s" % nv["build"]}
        except Exception as e:
            out["march_native"] = {"error": "%s: %s" % (type(e).__name__, e)}
    else:
        out["march_native"] = {"error": "no -march=native build on this host (g++ missing or failed)"}
    # the same oracle, frame-parallel over the host cores this job may use (one instance per thread;
    # the C call releases the interpreter lock)
    import threading
    ncores = max(1, min(16, len(os.sched_getaffinity(0))))
    counts = [0] * ncores
    budget = 4.0

    def worker(i):
        o = oracle_lib.Oracle(cfg["nfeat"], SCALE, cfg["nlevels"], INI_TH, MIN_TH)
        k = i
        while time.perf_counter() - t1 < budget:
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
                        "sample": "%d frames over %d threads in %.1f s" % (sum(counts), ncores, dt_all)}
    return out


def projection_stage(torch, oa, run, steps):
    """Times orbx_search_by_projection_batch_device on the batch the extractor just produced."""
    ex, n, cap, d_kps, d_desc, d_n = run.ex, run.n, run.cap, run.d_kps, run.d_desc, run.d_n
    W, H = run.cfg["w"], run.cfg["h"]
    fx, fy, cx, cy = 718.856, 718.856, 607.1928, 185.2157  # Examples/Stereo/KITTI00-02.yaml (rectified: no distortion)
    d_un = torch.zeros((n, cap, 28), dtype=torch.uint8, device="cuda")
    d_cs = torch.zeros((n, 64 * 48 + 1), dtype=torch.int32, device="cuda")
    d_ci = torch.zeros((n, cap), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()   # the fills above are on torch's stream; the kernel below writes the same buffers on the extractor's
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

    def run_once():
        ex.search_by_projection_batch_device(n, d_un.data_ptr(), cap, d_cs.data_ptr(), d_ci.data_ptr(), d_desc.data_ptr(), 0, 0,
                                             P, poses.data_ptr(), q.data_ptr(), cap, d_n.data_ptr(), d_match.data_ptr(),
                                             d_qm.data_ptr(), d_nm.data_ptr(), sync=False)
    ms = 1000.0 * timed_steps(run, steps, 1, run_once)
    return {"name": "ORBmatcher::SearchByProjection(current, last) for every frame of the batch, th 15, rotation check "
                    "(not part of value)",
            "ms_per_step": ms, "frames_per_s": n / (ms * 1e-3),
            "matched_fraction": float(d_nm.sum().item()) / max(1.0, float(d_n.sum().item()))}


def rate_entry(oa, cfg, n, sec_per_step, counts, note):
    kavg = float(np.mean(counts))
    b, _ = frame_bytes(oa, cfg, kavg)
    fps = n / sec_per_step
    return {"workload": note, "frames_per_step": n, "ms_per_step": 1000.0 * sec_per_step, "frames_per_s": fps,
            "keypoints_per_frame": kavg, "algorithmic_bytes_per_frame": b, "frac_of_hbm_peak": b * fps / 1e9 / HBM_PEAK_GBS}


def stereo_pair_latency(pairs=300):
    """One KITTI-shape stereo pair per call through the drop-in C++ class (tests/cpp/orbextractor_harness, built by
    __graft_entry__.build()): the reference's pattern -- two instances, two threads created per pair, then the
    association (src/Frame.cc:78-81, :90) -- beside the same on one thread and beside ExtractStereoPair, which is one
    captured graph launch on one instance.  Median milliseconds per pair, pyramids not copied back."""
    from orb_slam2_amd import synth
    exe = os.path.join(ROOT, "tests", "cpp", "orbextractor_harness")
    if not os.path.exists(exe):
        return {"error": "tests/cpp/orbextractor_harness not built"}
    w, h = C3["w"], C3["h"]
    base = synth.make_batch(8, w + 32, h, config_id=3, first_idx=900)
    fr = np.zeros((2 * pairs, h, w), np.uint8)
    for p in range(pairs):
        fr[2 * p], fr[2 * p + 1] = base[p % 8][:, :w], base[p % 8][:, 7:w + 7]
    out = {"workload": "1241x376, 2000 features, 8 levels: host images in, keypoints, descriptors, mvuRight and mvDepth out, "
                       "%d pairs, ORB_SLAM2::ORBextractor (C++ class) over the C ABI" % pairs}
    with tempfile.TemporaryDirectory(prefix="orbx_pair_") as d:
        raw = os.path.join(d, "pairs.raw")
        fr.tofile(raw)
        for key, mode in (("two_instances_two_threads", 1), ("two_instances_one_thread", 0), ("one_instance_one_graph", 2)):
            try:
                r = subprocess.run([exe, "threads", "2000", "1.2", "8", str(w), str(h), str(pairs), raw, "0.5371657", "386.1448",
                                    os.path.join(d, "o.bin"), str(mode)], capture_output=True, text=True, timeout=300,
                                   env=dict(os.environ, ORBX_KEEP_PYRAMID="0"))
                m = re.search(r"pair_ms median ([0-9.]+) p10 ([0-9.]+) p90 ([0-9.]+)", r.stdout)
                out[key] = ({"median": float(m.group(1)), "p10": float(m.group(2)), "p90": float(m.group(3))} if m else
                            {"error": (r.stderr or r.stdout)[-300:]})
            except Exception as e:
                out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def extras(torch, oa, device):
    """The other configurations of the benchmark suite on this GPU, each with its own algorithmic-byte fraction."""
    out = {}
    # ---- C2: 640x480 / 1000 features: single-frame latency of the host-memory entry point, batched rate ----
    try:
        from orb_slam2_amd import synth
        cfg = C2
        W, H = cfg["w"], cfg["h"]
        ex = oa.ORBextractor(cfg["nfeat"], SCALE, cfg["nlevels"], INI_TH, MIN_TH, device=device, max_width=W, max_height=H,
                             max_batch=1, keep_pyramid=False)
        L = ex._L
        imgs = [synth.make_frame(W, H, cfg["config_id"], i) for i in range(8)]
        cap = ex.max_keypoints(W, H)
        kps, desc, n = np.zeros(cap, oa.KP_DTYPE), np.zeros((cap, 32), np.uint8), C.c_int(0)
        pyr = [np.empty((g.height, g.width), np.uint8) for g in
               (oa.host_level_geometry(cfg["nfeat"], SCALE, cfg["nlevels"], W, H, l) for l in range(cfg["nlevels"]))]

        def call(i, with_pyramid):
            im = imgs[i % 8]
            rc = L.orbx_extract(ex._ctx, im.ctypes.data_as(C.c_void_p), W, H, im.strides[0], kps.ctypes.data_as(C.c_void_p),
                                cap, desc.ctypes.data_as(C.c_void_p), C.byref(n))
            assert rc == 0, rc
            if with_pyramid:   # what keeps mvImagePyramid valid for a caller that reads it (STEREO without the GPU stereo stage)
                for l, p in enumerate(pyr):
                    assert L.orbx_pyramid_level(ex._ctx, 0, l, p.ctypes.data_as(C.c_void_p), p.strides[0], None, None) == 0

        def call_padded(i, _):
            # mvImagePyramid for a host-side reader: the padded levels are built on the device inside the same submission
            # and read in place from the context's pinned memory (orbx_set_keep_pyramid / orbx_padded_level)
            call(i, False)
            base, lw, lh, st = C.c_void_p(), C.c_int(), C.c_int(), C.c_int()
            for l in range(cfg["nlevels"]):
                assert L.orbx_padded_level(ex._ctx, l, C.byref(base), C.byref(lw), C.byref(lh), C.byref(st)) == 0

        lat = {}
        for key, with_pyr, graph, reps in (("graph", False, 1, 1000), ("plain_launches", False, 0, 1000),
                                           ("graph_with_pyramid_readback", True, 1, 1000),
                                           ("graph_with_padded_pyramid", "padded", 1, 1000)):
            ex.set_debug_flag(oa.FLAG_NO_GRAPH, 0 if graph else 1)
            assert L.orbx_set_keep_pyramid(ex._ctx, 19 if with_pyr == "padded" else 0) == 0
            call_ = call_padded if with_pyr == "padded" else call
            for i in range(20):
                call_(i, with_pyr)
            t = []
            for i in range(reps):
                t0 = time.perf_counter()
                call_(i, with_pyr)
                t.append(time.perf_counter() - t0)
            t = np.array(t) * 1e3
            lat[key] = {"median": float(np.median(t)), "p10": float(np.percentile(t, 10)), "p90": float(np.percentile(t, 90)),
                        "calls": reps}
        ex.close()
        lat["stereo_pair"] = stereo_pair_latency()
        out["latency_ms"] = dict(lat, workload="C2: one 640x480 frame per orbx_extract call, 1000 features, 8 levels: host image "
                                               "in, host keypoints + descriptors out (upload, whole path, download); 'graph' = "
                                               "one HIP graph launch per call (the default), 'plain_launches' = the same "
                                               "launches issued one by one", keypoints=int(n.value))
        run = DeviceRun(torch, oa, cfg, 256, device, first_idx=1000)
        sec = timed_steps(run, 50, 3)
        out["c2_batched"] = rate_entry(oa, cfg, run.n, sec, run.d_n.cpu().numpy(),
                                       "C2: 256 frames of 640x480 per step resident in HBM, 1000 features, 8 levels")
        run.close()
        del run
    except Exception as e:
        out["c2_error"] = "%s: %s" % (type(e).__name__, e)
    # ---- C5: 3840x2160 / 8000 features / 12 levels ----
    try:
        run = DeviceRun(torch, oa, C5, 128, device, first_idx=2000, n_base=2)
        sec = timed_steps(run, 20, 3)   # enough steps that filling and draining the pipeline of batches does not show
        out["c5"] = rate_entry(oa, C5, run.n, sec, run.d_n.cpu().numpy(),
                               "C5: 128 frames of 3840x2160 per step resident in HBM, 8000 features, 12 levels")
        run.close()
        del run
    except Exception as e:
        out["c5"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # ---- C4 on this GPU alone: the 8192-frame EuRoC stream as 16 batches of 512 ----
    try:
        run = DeviceRun(torch, oa, C4, 512, device, first_idx=3000)
        nb = C4_STREAM // run.n

        def stream():
            for _ in range(nb):
                run.step()
        sec = timed_steps(run, 3, 1, stream)
        out["c4_one_gpu"] = rate_entry(oa, C4, C4_STREAM, sec, run.d_n.cpu().numpy(),
                                       "C4 on one GPU: 8192 EuRoC-shape frames (752x480, 1200 features, 8 levels) as 16 batches of "
                                       "512 resident in HBM; `--config c4` shards the stream over the ranks")
        run.close()
        del run
    except Exception as e:
        out["c4_one_gpu"] = {"error": "%s: %s" % (type(e).__name__, e)}
    torch.cuda.empty_cache()
    return out


def host_to_host(oa, run):
    """PCIe-inclusive rate of orbx_extract_batch: pageable host frames in, host keypoints/descriptors out."""
    n, cfg, cap, ex = run.n, run.cfg, run.cap, run.ex
    frames = [np.ascontiguousarray(run.base[i % len(run.base)]) for i in range(n)]
    kps, desc, cnt = np.zeros((n, cap), oa.KP_DTYPE), np.zeros((n, cap, 32), np.uint8), np.zeros(n, np.int32)
    ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
    strides = (C.c_int * n)(*[f.strides[0] for f in frames])
    args = (ex._ctx, n, ptrs, cfg["w"], cfg["h"], strides, kps.ctypes.data_as(C.c_void_p), cap,
            desc.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p))
    # The first calls of a process are not the steady state: the pinned slots are allocated in the first one, and in each
    # of the first three one hipMemcpyAsync of the download stream blocks the calling thread for about 6 ms inside the
    # runtime (profiles/r03_h2h_trace_*.txt).  They are timed, reported as first_calls_ms, and kept out of the median.
    first = []
    for _ in range(4):
        t0 = time.perf_counter()
        assert ex._L.orbx_extract_batch(*args) == 0
        first.append(time.perf_counter() - t0)
    t = []
    for _ in range(11):
        t0 = time.perf_counter()
        assert ex._L.orbx_extract_batch(*args) == 0
        t.append(time.perf_counter() - t0)
    med = float(np.median(t))
    e = rate_entry(oa, cfg, n, med, cnt, "orbx_extract_batch: %d KITTI-shape frames from pageable host memory, keypoints and "
                                         "descriptors back to host memory (PCIe both ways, chunked and overlapped)" % n)
    e["GBs_in"] = n * cfg["w"] * cfg["h"] / med / 1e9
    e["spread_ms"] = {"min": 1e3 * min(t), "max": 1e3 * max(t), "calls": len(t), "max_over_median": max(t) / med,
                      "calls_ms": [round(1e3 * v, 3) for v in t], "first_calls_ms": [round(1e3 * v, 3) for v in first]}
    return e


def host_stream(args):
    """`--config c4 --host`: the EuRoC-shape stream of 8192 frames from pageable HOST memory through the in-process
    worker pool (orbx_pool_*: one worker thread and one context per GPU, contiguous pair-aligned shards, pinned
    rings, results gathered in frame order in host memory).  One process, N devices; PCIe both ways is inside the
    timed region.  ORBX_POOL_DEVICES="0,0" lists the devices explicitly (a rehearsal with two workers on one GPU)."""
    import torch
    import orb_slam2_amd as oa
    from orb_slam2_amd import synth
    cfg = C4
    if os.environ.get("ORBX_POOL_DEVICES"):
        devices = [int(v) for v in os.environ["ORBX_POOL_DEVICES"].split(",")]
    else:
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit("--gpus %d but this node has %d device(s)" % (args.gpus, torch.cuda.device_count()))
        devices = list(range(args.gpus))
    W, H = cfg["w"], cfg["h"]
    base = synth.make_batch(64, W, H, config_id=cfg["config_id"], first_idx=3000)
    frames = [np.array(base[i % 64]) for i in range(C4_STREAM)]      # 8192 separate pageable arrays (3 GB)
    pool = oa.ExtractorPool(cfg["nfeat"], SCALE, cfg["nlevels"], INI_TH, MIN_TH, devices, max_width=W, max_height=H, max_batch=512)
    out = None
    for _ in range(max(1, min(args.warmup, 2))):
        out = pool.extract_stream(frames, out)
    t = []
    for _ in range(args.steps):
        t0 = time.perf_counter()
        out = pool.extract_stream(frames, out)
        t.append(time.perf_counter() - t0)
    dt = float(np.sum(t))
    counts = out[2]
    res = rate_entry(oa, cfg, C4_STREAM, dt / args.steps, counts, "EuRoC-shape stream")
    line = {"metric": "ORB frames/sec (752x480 EuRoC stream of 8192 frames, 1200 kp, 8 lvl), host frames in, host results out",
            "value": C4_STREAM * args.steps / dt, "unit": "frames/s", "n_gpus": len(set(devices)), "steps": args.steps,
            "warmup": max(1, min(args.warmup, 2)), "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "EuRoC-shape stereo stream, %d frames of 752x480 in pageable host memory, 1200 features/image, 8 "
                                   "levels, cut by frame over %d worker(s) on device(s) %s of ONE process; keypoints and descriptors "
                                   "back in host memory in frame order (PCIe both ways inside the timed region)"
                                   % (C4_STREAM, len(devices), devices),
                       "workers": len(devices), "devices": devices, "keypoints_per_frame": res["keypoints_per_frame"],
                       "parallelism": "in-process pool: one host worker thread and one context per GPU, no collective"},
            "spread_ms": {"min": 1e3 * min(t), "median": 1e3 * float(np.median(t)), "max": 1e3 * max(t)},
            "GBs_in": C4_STREAM * W * H * args.steps / dt / 1e9,
            "path_roofline": {"algorithmic_bytes_per_frame": res["algorithmic_bytes_per_frame"],
                              "frac_of_hbm_peak": res["frac_of_hbm_peak"] / max(1, len(set(devices)))},
            "cpu_baseline": None}
    pool.close()
    print(json.dumps(line))


def self_launch(n):
    """`bench.py --gpus N` started without a launcher (no WORLD_SIZE in the environment): starts the N ranks as fresh
    child processes -- one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set per child, the same command line --
    before anything in this process has touched a GPU, waits for them and returns the worst exit code.  Rank 0
    prints the JSON line.  Fewer than N devices is an error, not a smaller run (ORBX_BENCH_BACKEND=gloo, the
    rehearsal on a box with fewer GPUs than ranks, lets ranks share devices)."""
    import socket
    import torch   # device_count() does not initialise the GPU
    have = torch.cuda.device_count()
    if os.environ.get("ORBX_BENCH_BACKEND", "nccl") == "nccl" and have < n:
        sys.stderr.write("bench.py: --gpus %d but this node has %d device(s)\n" % (n, have))
        return 2
    if have < 1:
        sys.stderr.write("bench.py: no GPU\n")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst, live = 0, list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and not worst:
                worst = rc if rc > 0 else 1
                for q in live:   # a rank that died leaves the others in a barrier: end them (exact PIDs, started here)
                    q.terminate()
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pairs", type=int, default=PAIRS)
    ap.add_argument("--config", choices=("c2", "c3", "c4", "c5"), default="c3")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the other configurations and the consumers")
    ap.add_argument("--host", action="store_true",
                    help="with --config c4: host frames in, host results out, through the in-process worker pool over --gpus devices")
    args = ap.parse_args()
    if args.host:
        if args.config != "c4":
            raise SystemExit("--host goes with --config c4")
        if args.steps == 500:
            args.steps = 5      # a step is the whole 8192-frame stream over PCIe: about 0.1 s
        return host_stream(args)

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started without a launcher: this process starts the ranks itself and never touches a GPU
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    import orb_slam2_amd as oa
    from orb_slam2_amd import shard

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # never a line that reports fewer GPUs than were asked for
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("ORBX_BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < (local_rank + 1 if world > 1 else 1):
        raise SystemExit("rank %d needs device %d, the node has %d" % (rank, local_rank, torch.cuda.device_count()))
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

    def barrier():
        if world > 1:
            dist.barrier()

    global COUNTER_PREFIX
    if args.config == "c3":
        cfg, n = C3, 2 * args.pairs
        run = DeviceRun(torch, oa, cfg, n, local_rank, first_idx=rank * ((n + 7) // 8))
        job_frames = world * n        # weak scaling: every rank its own batch
        rank_frames = n
        step = run.step
        scaling = "weak"
        workload = ("KITTI stereo 1241x376 left+right, 2000 features/image, 8 levels, scale 1.2, FAST 20/7, %d pairs (%d frames) "
                    "per step per GPU, frames resident in HBM" % (args.pairs, n))
    elif args.config == "c2":
        # the TUM configuration batched, as the measured step (profiles of its kernels); `extras` reports the same step
        cfg, n = C2, min(256, 2 * args.pairs)
        COUNTER_PREFIX = "c2_"
        PROFILED_FRAMES_NOW[0] = PROFILED_FRAMES["c2"]
        run = DeviceRun(torch, oa, cfg, n, local_rank, first_idx=1000 + rank * ((n + 7) // 8))
        job_frames = world * n
        rank_frames = n
        step = run.step
        scaling = "weak"
        workload = ("TUM 640x480, 1000 features/image, 8 levels, scale 1.2, FAST 20/7, %d frames per step per GPU, frames resident "
                    "in HBM" % n)
    elif args.config == "c5":
        COUNTER_PREFIX = "c5_"
        PROFILED_FRAMES_NOW[0] = PROFILED_FRAMES["c5"]
        # the 4K configuration as the measured step (profiles of its kernels); `extras` reports the same step
        cfg, n = C5, min(128, 2 * args.pairs)
        run = DeviceRun(torch, oa, cfg, n, local_rank, first_idx=2000 + rank, n_base=2)
        job_frames = world * n
        rank_frames = n
        step = run.step
        scaling = "weak"
        workload = ("3840x2160, 8000 features/image, 12 levels, scale 1.2, FAST 20/7, %d frames per step per GPU, frames "
                    "resident in HBM" % n)
    else:
        cfg = C4
        first, last = shard.frame_shard(C4_STREAM, rank, world)   # strong scaling: the stream is cut over the ranks
        n = min(512, last - first)
        run = DeviceRun(torch, oa, cfg, n, local_rank, first_idx=first // 8)
        chunks = [min(n, last - f) for f in range(first, last, n)]
        job_frames = C4_STREAM
        rank_frames = last - first

        def step():
            for cnt in chunks:
                run.step(0, cnt)
        scaling = "strong"
        workload = ("EuRoC-shape stereo stream, %d frames of 752x480, 1200 features/image, 8 levels, cut by frame over %d rank(s), "
                    "each rank in batches of %d resident in HBM; one step = the whole stream" % (C4_STREAM, world, n))
    ex, cap = run.ex, run.cap

    for _ in range(args.warmup):
        step()
    ex.synchronize()
    ex.set_profiling(True)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()          # steps are enqueued back to back: a batch's descriptors overlap the next batch's FAST
    ex.synchronize()    # the extractor's own streams (torch's current stream never sees them)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = shard.max_over_ranks(time.perf_counter() - t0, dist, red_dev)  # the job is as slow as its slowest rank
    # HIP events recorded on the extractor's streams around each stage's launches: the events of the LAST batch
    # of the timed region are the ones left to read
    stage_ms = ex.stage_ms()

    # Spread of the step time: further blocks of steps, each enqueued back to back and timed on its own.
    blk = max(1, min(args.steps, 100))
    blocks = []
    t_all = time.perf_counter()
    while len(blocks) < 5 or (time.perf_counter() - t_all < 1.0 and len(blocks) < 50):
        blocks.append(1000.0 * timed_steps(run, blk, 0, step))

    # Outside the timed region: the same steps with the stages serialised on one stream, so that each
    # kernel's duration is its own (in the timed region the pyramid and the blur run beside FAST and the
    # quadtree on other streams and stretch each other).
    serial_sum = {}
    ex.set_debug_flag(oa.FLAG_NO_OVERLAP, 1)
    run.step()
    ex.synchronize()
    for _ in range(5):
        run.step()
        ex.synchronize()
        for k, v in ex.stage_ms().items():
            serial_sum[k] = serial_sum.get(k, 0.0) + v / 5
    ex.set_debug_flag(oa.FLAG_NO_OVERLAP, 0)
    ex.set_profiling(False)

    next_stage = proj = h2h = None
    if rank == 0 and args.config == "c3" and not args.no_extras:
        # the stage after the extractor (stereo association of the pairs of the batch), timed on its own:
        # it is not part of `value`
        run.step()
        ex.synchronize()
        d_u = torch.empty((args.pairs, cap), dtype=torch.float32, device="cuda")
        d_dep = torch.empty((args.pairs, cap), dtype=torch.float32, device="cuda")
        mb, mbf = 0.5371657, 386.1448  # KITTI 00-02
        torch.cuda.synchronize()

        def stereo():
            ex.stereo_match_batch_device(args.pairs, run.d_kps.data_ptr(), cap, run.d_desc.data_ptr(), run.d_n.data_ptr(), mb,
                                         mbf, d_u.data_ptr(), d_dep.data_ptr(), sync=False)
        k_cons = max(1, min(args.steps, 50))
        stereo_ms = 1000.0 * timed_steps(run, k_cons, 1, stereo)
        counts = run.d_n.cpu().numpy()
        next_stage = {"name": "stereo association of the batch's pairs (not part of value)", "ms_per_step": stereo_ms,
                      "pairs_per_s": args.pairs / (stereo_ms * 1e-3),
                      "matched_fraction": float((d_u >= 0).float().sum().item() / max(1.0, float(counts[0::2].sum())))}
        # a descriptor consumer further down the tracking thread, timed on its own (not part of `value`):
        # ORBmatcher::SearchByProjection between consecutive frames, every frame given its own features back
        # as the last frame's map points, seen from a pose 30 cm further on
        try:
            proj = projection_stage(torch, oa, run, k_cons)
        except Exception as e:  # the headline measurement does not depend on this stage
            proj = {"error": "%s: %s" % (type(e).__name__, e)}
        del d_u, d_dep
        if world == 1:
            try:
                h2h = host_to_host(oa, run)
            except Exception as e:
                h2h = {"error": "%s: %s" % (type(e).__name__, e)}

    out = None
    if rank == 0:
        counts = run.d_n.cpu().numpy()
        kavg = float(counts.mean())
        b_frame, px = frame_bytes(oa, cfg, kavg)
        P = sum(px)
        fps = job_frames * args.steps / dt
        # compulsory bytes per frame of each stage (read once, write once)
        stage_bytes = {"pyramid": (P - px[-1]) + (P - px[0]), "fast": P,
                       "quadtree": 0,  # candidate lists: outside the algorithmic byte count
                       "blur": 2 * P, "orient_describe": 1321 * kavg}
        dom = max((k for k in stage_ms if stage_bytes[k] > 0), key=lambda k: stage_ms[k])
        launches = {"pyramid": cfg["nlevels"] - 1, "fast": 1, "quadtree": 1, "blur": 1, "orient_describe": 1}
        # per launch: algorithmic bytes of the stage / its launches, over the average launch duration
        dom_bytes = stage_bytes[dom] * n / launches[dom]
        achieved = dom_bytes / (stage_ms[dom] / launches[dom] * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(dom, n, launches[dom])
        roof = {
            "bound": "hbm", "kernel": dom, "launches_per_step": launches[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": dom_bytes, "kernel_ms_per_launch": stage_ms[dom] / launches[dom],
            "note": "stage times are those of the last batch of the timed region: behind another batch FAST runs as 1 launch "
                    "(its pyramid was built while the batch before was detected) with the next batch's pyramid and blur, this "
                    "batch's quadtree and the previous batch's descriptors beside it on three other streams; serialised_* are "
                    "the same kernels alone on the GPU (measured after the timed region)",
            "serialised_ms_per_launch": serial_sum[dom] / launches[dom],
            "serialised_frac": stage_bytes[dom] * n / (serial_sum[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "valu_issue": valu_view(dom, n, serial_sum[dom]),
        }
        out = {
            "metric": {"c3": "ORB frames/sec (1241x376, 2000 kp, 8 lvl)",
                       "c4": "ORB frames/sec (752x480 EuRoC stream of 8192 frames, 1200 kp, 8 lvl)",
                       "c2": "ORB frames/sec (640x480, 1000 kp, 8 lvl)",
                       "c5": "ORB frames/sec (3840x2160, 8000 kp, 12 lvl)"}[args.config],
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "frames_per_step_per_gpu": rank_frames, "keypoints_per_frame": kavg,
                       "parallelism": "frames sharded across GPUs, no collective"},
            "ms_per_step_blocks": {"median": float(np.median(blocks)), "min": min(blocks), "max": max(blocks),
                                   "blocks": len(blocks), "steps_per_block": blk,
                                   "note": "rank 0, after the timed region: each block enqueued back to back and synchronised"},
            "roofline": roof,
            "path_roofline": {"algorithmic_bytes_per_frame": b_frame, "achieved_GBs_per_gpu": b_frame * fps / world / 1e9,
                              "frac_of_hbm_peak": b_frame * fps / world / 1e9 / HBM_PEAK_GBS},
            "path_valu_issue": path_valu_view(n, 1000.0 * dt / args.steps, PROFILED_FRAMES[args.config]),
            "stage_ms": stage_ms, "stage_ms_serialised": serial_sum,
            "next_stage": next_stage, "next_stage_matching": proj, "host_to_host": h2h,
        }
    base = run.base
    run.close()
    del run
    torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and args.config == "c3" and not args.no_extras:
            out.update(extras(torch, oa, local_rank))
        # the CPU leg runs on rank 0 at every N, on the frames the GPU leg was generated from
        out["cpu_baseline"] = None if args.no_cpu else cpu_baseline(base, cfg)
        print(json.dumps(out))
    barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
