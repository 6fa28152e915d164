"""GPU parity of the matching loops of ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist)
(reference src/ORBmatcher.cc:1509-1636) and SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (:290-403) on projected
map points against the CPU oracle's sequential loop, bit-exact: the point each feature receives, the feature each point
is written to, and nmatches."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth
from hip_rt import DeviceArray

pytestmark = pytest.mark.gpu


def run_device(ex, un, cs, ci, desc, taken, grid, queries, nq, level_up, max_dist, check):
    n, kp_cap, q_cap = un.shape[0], un.shape[1], queries.shape[1]
    d = dict(un=DeviceArray.from_host(un.view(np.uint8)), cs=DeviceArray.from_host(cs), ci=DeviceArray.from_host(ci),
             desc=DeviceArray.from_host(desc), q=DeviceArray.from_host(queries.view(np.uint8)), nq=DeviceArray.from_host(nq),
             match=DeviceArray((n, kp_cap), np.int32, fill_byte=0x55), qm=DeviceArray((n, q_cap), np.int32, fill_byte=0x55),
             nm=DeviceArray((n,), np.int32, fill_byte=0x55))
    if taken is not None:
        d["tk"] = DeviceArray.from_host(taken)
    ex.search_guided_batch_device(n, d["un"].ptr, kp_cap, d["cs"].ptr, d["ci"].ptr, d["desc"].ptr, d["tk"].ptr if "tk" in d else 0,
                                  grid[0], grid[1], grid[2], grid[3], d["q"].ptr, q_cap, d["nq"].ptr, level_up, max_dist, check,
                                  d["match"].ptr, d["qm"].ptr, d["nm"].ptr, sync=True)
    out = d["match"].to_host(), d["qm"].to_host(), d["nm"].to_host()
    for a in d.values():
        a.free()
    return out


def check_against_oracle(oracle, un, counts, cs, ci, desc, taken, grid, queries, nq, level_up, max_dist, check, match, qm, nm):
    total = 0
    for f in range(un.shape[0]):
        k = int(counts[f])
        xy = np.stack([un[f, :k]["x"], un[f, :k]["y"]], 1)
        on, om, oqm = oracle.search_guided(xy, un[f, :k]["octave"], un[f, :k]["angle"], desc[f, :k],
                                           None if taken is None else taken[f, :k], cs[f], ci[f, :k], grid[0], grid[1], grid[2], grid[3],
                                           queries[f, :nq[f]], level_up, max_dist, check)
        assert nm[f] == on, (f, nm[f], on)
        assert (qm[f, :nq[f]] == oqm).all(), f
        assert (match[f, :k] == om).all(), f
        total += on
    return total


@pytest.mark.parametrize("level_up,th,max_dist,check,use_taken", [(1, 10.0, 100, True, True), (1, 3.0, 64, True, False),
                                                                 (0, 10.0, 50, False, True), (0, 4.0, 50, False, False)])
def test_guided_search_after_extraction(oracle, level_up, th, max_dist, check, use_taken):
    """Real extractions; the map points are the keypoints of the next view of the scene (shifted by a few pixels) projected
    a little off, in index order, some twice, some skipped; a third of the features hold a map point already."""
    w, h, nf, nl, n = 640, 480, 1000, 8, 3
    base = synth.make_frame(w + 16, h + 16, 2, 77)
    frames = np.stack([np.ascontiguousarray(base[2 * i:2 * i + h, 3 * i:3 * i + w]) for i in range(n)])
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
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
    rng = np.random.default_rng(int(th) + 100 * level_up)
    scale = ex.GetScaleFactors()
    q_cap = cap + 100
    queries = np.zeros((n, q_cap), oa.GUIDED_QUERY_DTYPE)
    nq = np.zeros(n, np.int32)
    taken = (rng.random((n, cap)) < 0.3).astype(np.uint8) if use_taken else None
    for f in range(n):
        o = (f + 1) % n                                   # the keyframe whose map points are projected into frame f
        k = int(counts[o])
        pick = np.concatenate([np.arange(k), rng.choice(k, 100)])[:q_cap]
        src = un[o, pick]
        m = len(pick)
        q = queries[f, :m]
        shift = np.array([3.0 * (o - f), 2.0 * (o - f)], np.float32)
        q["u"] = src["x"] + shift[0] + rng.normal(0, 1.5, m).astype(np.float32)
        q["v"] = src["y"] + shift[1] + rng.normal(0, 1.5, m).astype(np.float32)
        q["angle"] = src["angle"]
        q["level"] = np.clip(src["octave"] + rng.integers(-1, 2, m), 0, nl - 1)
        q["radius"] = np.float32(th) * scale[q["level"]]
        q["flags"] = rng.choice(np.array([1, 1, 1, 1, 1, 3], np.int32), m)
        noisy = desc[o, pick].copy()
        flip = rng.random((m, 32)) < 0.06
        noisy[flip] ^= (1 << rng.integers(0, 8, (m, 32)).astype(np.uint8))[flip]
        q["desc"] = noisy
        nq[f] = m
    match, qm, nm = run_device(ex, un, cs, ci, desc, taken, grid, queries, nq, level_up, max_dist, check)
    total = check_against_oracle(oracle, un, counts, cs, ci, desc, taken, grid, queries, nq, level_up, max_dist, check, match, qm, nm)
    assert total > 150 * n
    ex.close()
    for a in (d_img, d_kps, d_desc, d_n, d_un, d_cs, d_ci):
        a.free()


def test_guided_search_crowds_and_arguments(oracle):
    """Hundreds of points aimed at one spot with near-identical descriptors: long chains of points that lose their best
    feature to an earlier one."""
    ex = oa.ORBextractor(500, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=2, keep_pyramid=False)
    rng = np.random.default_rng(5)
    grid = (0.0, 0.0, 0.1, 0.1)
    n, k, q_cap = 2, 400, 600
    un = np.zeros((n, k), oa.KP_DTYPE)
    desc = np.zeros((n, k, 32), np.uint8)
    cs, ci = np.zeros((n, 64 * 48 + 1), np.int32), np.zeros((n, k), np.int32)
    queries = np.zeros((n, q_cap), oa.GUIDED_QUERY_DTYPE)
    nq = np.array([q_cap, 350], np.int32)
    for f in range(n):
        un[f]["x"] = 300 + rng.uniform(-6, 6, k)
        un[f]["y"] = 200 + rng.uniform(-6, 6, k)
        un[f]["octave"] = rng.integers(0, 3, k)
        un[f]["angle"] = rng.choice(np.array([0.0, 10.0, 45.0, 200.0], np.float32), k)
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        desc[f] = base
        flip = rng.random((k, 32)) < 0.1
        desc[f][flip] ^= (1 << rng.integers(0, 8, (k, 32)).astype(np.uint8))[flip]
        cs[f], idx = oracle.assign_grid(np.stack([un[f]["x"], un[f]["y"]], 1), *grid)
        ci[f, :len(idx)] = idx
        q = queries[f]
        q["u"], q["v"] = 300 + rng.uniform(-3, 3, q_cap), 200 + rng.uniform(-3, 3, q_cap)
        q["level"] = rng.integers(0, 3, q_cap)
        q["radius"] = rng.choice(np.array([3.0, 6.0, 12.0], np.float32), q_cap)
        q["angle"] = rng.choice(np.array([0.0, 10.0, 45.0, 200.0], np.float32), q_cap)
        q["flags"] = 1
        q["desc"] = base
        flipq = rng.random((q_cap, 32)) < 0.1
        q["desc"][flipq] ^= (1 << rng.integers(0, 8, (q_cap, 32)).astype(np.uint8))[flipq]
    counts = np.array([k, k])
    for level_up, max_dist, check in ((1, 100, True), (0, 50, False), (1, 30, True)):
        match, qm, nm = run_device(ex, un, cs, ci, desc, None, grid, queries, nq, level_up, max_dist, check)
        total = check_against_oracle(oracle, un, counts, cs, ci, desc, None, grid, queries, nq, level_up, max_dist, check, match, qm, nm)
        assert total > 100
    with pytest.raises(oa.OrbxError):
        ex.search_guided_batch_device(1, 16, 100, 16, 16, 16, 0, 0, 0, 0.1, 0.1, 16, 100, 16, 2, 50, 0, 16, 16, 16)   # level_up above 1
    with pytest.raises(oa.OrbxError):
        ex.search_guided_batch_device(1, 16, 100, 16, 16, 16, 0, 0, 0, 0.1, 0.1, 0, 100, 16, 1, 50, 0, 16, 16, 16)    # null queries
    ex.close()
