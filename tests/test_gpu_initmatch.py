"""GPU parity of ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:418-549) against the CPU oracle's
sequential loop, bit-exact: vnMatches12, nmatches and the updated vbPrevMatched, on the chain extract -> undistort +
grid -> match between shifted views of one scene, on crowds (many keypoints wanting the same few features, so that
matches change hands again and again) and on the oracle's hand-derived cases."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth
from hip_rt import DeviceArray

pytestmark = pytest.mark.gpu


def run_device(ex, un1, desc1, n1, prev, un2, cs2, ci2, desc2, grid, window, ratio, check):
    n, cap_1, cap_2 = un1.shape[0], un1.shape[1], un2.shape[1]
    d = [DeviceArray.from_host(np.ascontiguousarray(a)) for a in
         (un1.view(np.uint8), desc1, n1, prev, un2.view(np.uint8), cs2, ci2, desc2)]
    d_match = DeviceArray((n, cap_1), np.int32, fill_byte=0x55)
    d_nm = DeviceArray((n,), np.int32, fill_byte=0x55)
    ex.search_for_initialization_batch_device(n, d[0].ptr, d[1].ptr, d[2].ptr, cap_1, d[3].ptr, d[4].ptr, d[5].ptr, d[6].ptr,
                                              d[7].ptr, cap_2, grid[0], grid[1], grid[2], grid[3], window, ratio, check,
                                              d_match.ptr, d_nm.ptr, sync=True)
    out = d_match.to_host(), d_nm.to_host(), d[3].to_host()
    for a in d + [d_match, d_nm]:
        a.free()
    return out


def check_against_oracle(oracle, un1, desc1, n1, prev, un2, n2, cs2, ci2, desc2, grid, window, ratio, check, match, nm, prev_out):
    total = 0
    for p in range(un1.shape[0]):
        k1, k2 = int(n1[p]), int(n2[p])
        a, b = un1[p, :k1], un2[p, :k2]
        exp_n, exp_m, exp_pm = oracle.search_for_initialization(
            a["octave"], a["angle"], desc1[p, :k1], prev[p, :k1], np.stack([b["x"], b["y"]], 1), b["octave"], b["angle"],
            desc2[p, :k2], cs2[p], ci2[p, :k2], grid[0], grid[1], grid[2], grid[3], window, ratio, check)
        assert nm[p] == exp_n, "nmatches of pair %d: %d, expected %d" % (p, nm[p], exp_n)
        assert (match[p, :k1] == exp_m).all(), "matches of pair %d" % p
        assert (match[p, k1:] == -1).all()
        assert (prev_out[p, :k1] == exp_pm).all(), "vbPrevMatched of pair %d" % p
        assert (prev_out[p, k1:] == prev[p, k1:]).all()
        total += exp_n
    return total


@pytest.mark.parametrize("window,ratio,check,list_cap", [(100, 0.9, True, 0), (100, 0.9, False, 8), (30, 0.8, True, 0),
                                                        (250, 0.95, True, 3)])
def test_initialization_match_after_extraction(oracle, window, ratio, check, list_cap):
    """Four views of one scene, shifted by a few pixels; pair p matches view p with view p + 1.  vbPrevMatched starts at
    the first view's own keypoints (Tracking.cc:632-634) for two pairs and somewhere else for the others.  list_cap > 0
    makes the candidate lists overflow, so that the windows are walked a second time."""
    w, h, nf, n = 640, 480, 1000, 4
    base = synth.make_frame(w + 32, h + 32, 2, 41)
    frames = np.stack([np.ascontiguousarray(base[3 * i:3 * i + h, 5 * i:5 * i + w]) for i in range(n)])
    ex = oa.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
    ex.set_debug_flag(oa.FLAG_INIT_LIST, list_cap)
    cap = ex.max_keypoints(w, h)
    d_img = DeviceArray.from_host(frames)
    d_kps, d_desc = DeviceArray((n, cap, 28), np.uint8, fill_byte=0), DeviceArray((n, cap, 32), np.uint8, fill_byte=0)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    d_un = DeviceArray((n, cap, 28), np.uint8, fill_byte=0)
    d_cs, d_ci = DeviceArray((n, 64 * 48 + 1), np.int32, fill_byte=0xFF), DeviceArray((n, cap), np.int32, fill_byte=0xFF)
    grid = (0.0, 0.0, float(np.float32(64) / np.float32(w)), float(np.float32(48) / np.float32(h)))
    ex.extract_batch_device(d_img.ptr, n, w, h, w, h * w, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=False)
    ex.undistort_grid_batch_device(n, d_kps.ptr, cap, d_n.ptr, np.array([500, 500, 320, 240], np.float32), np.zeros(5, np.float32),
                                   grid[0], grid[1], grid[2], grid[3], d_un.ptr, d_cs.ptr, d_ci.ptr, sync=True)
    counts = d_n.to_host()
    un = d_un.to_host().reshape(n, cap * 28).view(oa.KP_DTYPE).reshape(n, cap)
    desc, cs, ci = d_desc.to_host(), d_cs.to_host(), d_ci.to_host()
    nxt = np.roll(np.arange(n), -1)
    rng = np.random.default_rng(window)
    prev = np.stack([un["x"], un["y"]], 2).astype(np.float32)
    prev[2:] += rng.normal(0, 20, prev[2:].shape).astype(np.float32)
    match, nm, prev_out = run_device(ex, un, desc, counts, prev, un[nxt], cs[nxt], ci[nxt], desc[nxt], grid, window, ratio, check)
    total = check_against_oracle(oracle, un, desc, counts, prev, un[nxt], counts[nxt], cs[nxt], ci[nxt], desc[nxt], grid, window,
                                 ratio, check, match, nm, prev_out)
    assert total > 40 * n                 # the views do share level-0 features
    # a second round from the updated positions, as Tracking::MonocularInitialization does frame after frame
    match2, nm2, prev_out2 = run_device(ex, un, desc, counts, prev_out, un[nxt], cs[nxt], ci[nxt], desc[nxt], grid, window, ratio,
                                        check)
    check_against_oracle(oracle, un, desc, counts, prev_out, un[nxt], counts[nxt], cs[nxt], ci[nxt], desc[nxt], grid, window, ratio,
                         check, match2, nm2, prev_out2)
    ex.close()
    for a in (d_img, d_kps, d_desc, d_n, d_un, d_cs, d_ci):
        a.free()


@pytest.mark.parametrize("k1,k2,cap,list_cap,seed", [(900, 40, 1024, 0, 1), (3000, 700, 4096, 0, 2), (8192, 8192, 8192, 0, 3),
                                                     (500, 300, 512, 5, 4), (64, 1, 64, 0, 5)])
def test_initialization_match_crowds(oracle, k1, k2, cap, list_cap, seed):
    """Near-identical descriptors in a few clusters: most keypoints of the first frame want the same features of the
    second, better claims arrive late, and every window holds hundreds of candidates (lists overflow)."""
    rng = np.random.default_rng(seed)
    ex = oa.ORBextractor(500, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=2, keep_pyramid=False)
    ex.set_debug_flag(oa.FLAG_INIT_LIST, list_cap)
    n = 2
    grid = (0.0, 0.0, 0.1, 0.1)
    un1, un2 = np.zeros((n, cap), oa.KP_DTYPE), np.zeros((n, cap), oa.KP_DTYPE)
    desc1 = rng.integers(0, 256, (n, cap, 32), dtype=np.uint8)
    desc2 = rng.integers(0, 256, (n, cap, 32), dtype=np.uint8)
    cs2, ci2 = np.zeros((n, 64 * 48 + 1), np.int32), np.zeros((n, cap), np.int32)
    n1, n2 = np.array([k1, max(1, k1 - 37)], np.int32), np.array([k2, max(1, k2 - 5)], np.int32)
    angles = np.array([0.0, 1.5, 29.9, 45.0, 90.0, 200.0, 359.5], np.float32)
    for p in range(n):
        spread = 40.0 if p == 0 else 250.0
        base = rng.integers(0, 256, (4, 32), dtype=np.uint8)
        for un, desc, k in ((un1, desc1, n1[p]), (un2, desc2, n2[p])):
            un[p, :k]["x"] = 320 + rng.uniform(-spread, spread, k)
            un[p, :k]["y"] = 240 + rng.uniform(-spread * 0.7, spread * 0.7, k)
            un[p, :k]["octave"] = rng.choice(np.array([0, 0, 0, 1, 2], np.int32), k)
            un[p, :k]["angle"] = rng.choice(angles, k)
            d = base[rng.integers(0, 4, k)]
            flip = rng.random((k, 32)) < 0.05
            d[flip] ^= (1 << rng.integers(0, 8, (k, 32)).astype(np.uint8))[flip]
            desc[p, :k] = d
        k = int(n2[p])
        cs2[p], idx = oracle.assign_grid(np.stack([un2[p, :k]["x"], un2[p, :k]["y"]], 1), *grid)
        ci2[p, :len(idx)] = idx
    prev = np.stack([un1["x"], un1["y"]], 2).astype(np.float32)
    for window, ratio, check in ((100, 0.9, True), (15, 0.99, False)):
        match, nm, prev_out = run_device(ex, un1, desc1, n1, prev, un2, cs2, ci2, desc2, grid, window, ratio, check)
        total = check_against_oracle(oracle, un1, desc1, n1, prev, un2, n2, cs2, ci2, desc2, grid, window, ratio, check, match, nm,
                                     prev_out)
        assert total >= 1
    ex.close()


def test_initialization_match_hand_cases_and_arguments(oracle):
    """The replaced-entry histogram case of the oracle's known-answer tests through the device path, an empty
    second frame, and argument checks."""
    ex = oa.ORBextractor(500, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1, keep_pyramid=False)
    grid = (0.0, 0.0, 0.1, 0.1)
    ones = lambda k: np.packbits(np.arange(256) < k)
    f1 = [(50 + 40 * i, 50, 0, 0.0) for i in range(10)] + [(100, 300, 0, 90.0), (200, 300, 0, 180.0), (300, 300, 20, 270.0),
                                                            (301, 300, 5, 270.0)]
    f2 = [(50 + 40 * i, 50, 0, 0.0) for i in range(10)] + [(100, 300, 0, 0.0), (200, 300, 0, 0.0), (300, 300, 0, 0.0)]
    cap = 16
    un1, un2 = np.zeros((1, cap), oa.KP_DTYPE), np.zeros((1, cap), oa.KP_DTYPE)
    desc1, desc2 = np.zeros((1, cap, 32), np.uint8), np.zeros((1, cap, 32), np.uint8)
    for un, desc, feats in ((un1, desc1, f1), (un2, desc2, f2)):
        for i, (x, y, bits, ang) in enumerate(feats):
            un[0, i]["x"], un[0, i]["y"], un[0, i]["angle"] = x, y, ang
            desc[0, i] = ones(bits)
    cs2, ci2 = np.zeros((1, 64 * 48 + 1), np.int32), np.zeros((1, cap), np.int32)
    cs2[0], idx = oracle.assign_grid(np.stack([un2[0, :len(f2)]["x"], un2[0, :len(f2)]["y"]], 1), *grid)
    ci2[0, :len(idx)] = idx
    prev = np.stack([un1["x"], un1["y"]], 2).astype(np.float32)
    n1 = np.array([len(f1)], np.int32)
    match, nm, prev_out = run_device(ex, un1, desc1, n1, prev, un2, cs2, ci2, desc2, grid, 10, 0.9, True)
    assert nm[0] == 12 and list(match[0, 10:14]) == [10, -1, -1, 12] and list(match[0, :10]) == list(range(10))
    assert prev_out[0, 13].tolist() == [300.0, 300.0] and prev_out[0, 12].tolist() == [300.0, 300.0]
    # an empty second frame: every cell is empty
    cs0 = np.zeros((1, 64 * 48 + 1), np.int32)
    match, nm, prev_out = run_device(ex, un1, desc1, n1, prev, un2, cs0, ci2, desc2, grid, 100, 0.9, True)
    assert nm[0] == 0 and (match == -1).all() and (prev_out == prev).all()
    with pytest.raises(oa.OrbxError):
        ex.search_for_initialization_batch_device(1, 16, 16, 16, 9000, 16, 16, 16, 16, 16, 100, 0, 0, 0.1, 0.1, 100, 0.9, 1, 16, 16)
    with pytest.raises(oa.OrbxError):
        ex.search_for_initialization_batch_device(1, 16, 16, 16, 100, 0, 16, 16, 16, 16, 100, 0, 0, 0.1, 0.1, 100, 0.9, 1, 16, 16)
    ex.close()
