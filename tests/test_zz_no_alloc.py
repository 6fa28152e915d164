"""The entry point that allocates nothing (orbx_stereo_match_batch_device_noalloc), through the C ABI,
against the entry point it is a variant of (orbx_stereo_match_batch_device): bit-exact on two calls
in a row, the second with a smaller input; then the size limit; then against the CPU oracle on a
context that never ran the allocating variant, with the device's free memory unchanged across the
call; then two contexts on one GPU using the one scratch array from two threads."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth

pytestmark = pytest.mark.gpu
MB, MBF = 0.11, 47.9


def _pair(w, h, cid, idx, shift):
    wide = synth.make_frame(w + 32, h, cid, idx)
    left = np.ascontiguousarray(wide[:, :w])
    right = np.clip(wide[:, shift:w + shift].astype(np.int32) +
                    np.random.default_rng(idx).integers(-3, 4, (h, w)), 0, 255).astype(np.uint8)
    return left, np.ascontiguousarray(right)


def _run_both(ex, w, h, n_pairs, first_idx):
    """Extracts n_pairs fresh pairs, runs both stereo entry points on them, returns their outputs."""
    from hip_rt import DeviceArray
    n = 2 * n_pairs
    stride = (w + 63) // 64 * 64
    host = np.zeros((n, h, stride), np.uint8)
    for p in range(n_pairs):
        l, r = _pair(w, h, 4, first_idx + p, 4 + 2 * p)
        host[2 * p, :, :w], host[2 * p + 1, :, :w] = l, r
    frames = DeviceArray.from_host(host)
    cap = ex.max_keypoints(w, h)
    d_kps, d_desc = DeviceArray((n, cap, 28), np.uint8), DeviceArray((n, cap, 32), np.uint8)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    outs = [(DeviceArray((n_pairs, cap), np.float32, fill_byte=0x41), DeviceArray((n_pairs, cap), np.float32, fill_byte=0x41))
            for _ in range(2)]
    ex.extract_batch_device(frames.ptr, n, w, h, stride, h * stride, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=False)
    ex.stereo_match_batch_device(n_pairs, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, MB, MBF, outs[0][0].ptr, outs[0][1].ptr,
                                 sync=True)
    ex.stereo_match_batch_device_noalloc(n_pairs, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, MB, MBF, outs[1][0].ptr,
                                         outs[1][1].ptr, sync=True)
    counts = d_n.to_host()
    res = [(u.to_host(), d.to_host()) for u, d in outs]
    for a in (frames, d_kps, d_desc, d_n, outs[0][0], outs[0][1], outs[1][0], outs[1][1]):
        a.free()
    return counts, res


def test_noalloc_equals_allocating_entry_point():
    w, h, nf, nl = 752, 480, 1200, 8
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=12, keep_pyramid=False)
    for n_pairs, first in ((6, 100), (2, 200)):          # the second call is smaller: it sees the first one's leftovers
        counts, ((u0, d0), (u1, d1)) = _run_both(ex, w, h, n_pairs, first)
        matched = 0
        for p in range(n_pairs):
            k = counts[2 * p]
            assert k > 500
            assert (u0[p, :k].view(np.uint32) == u1[p, :k].view(np.uint32)).all(), "pair %d" % p
            assert (d0[p, :k].view(np.uint32) == d1[p, :k].view(np.uint32)).all(), "pair %d" % p
            matched += int((u1[p, :k] >= 0).sum())
        assert matched > 100 * n_pairs
    ex.close()


def test_noalloc_refuses_sizes_beyond_its_scratch():
    from hip_rt import DeviceArray
    w, h, nf, nl = 640, 480, 1000, 8
    n = 4
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    host = np.zeros((n, h, w), np.uint8)
    host[:] = synth.make_batch(n, w, h, 1, 300)
    frames = DeviceArray.from_host(host)
    # a caller may size its arrays far above the extractor's own bound; 16 * kp_cap bytes per pair are scratch
    big = oa.NOALLOC_SCRATCH_BYTES // (16 * 2) + 1024
    big = min(big, 16384)
    d_kps, d_desc = DeviceArray((n, big, 28), np.uint8), DeviceArray((n, big, 32), np.uint8)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    d_u, d_d = DeviceArray((n // 2, big), np.float32), DeviceArray((n // 2, big), np.float32)
    ex.extract_batch_device(frames.ptr, n, w, h, w, h * w, d_kps.ptr, big, d_desc.ptr, d_n.ptr, sync=True)
    assert 2 * 16 * big < oa.NOALLOC_SCRATCH_BYTES       # kp_cap 16384: 2 pairs still fit
    ex.stereo_match_batch_device_noalloc(2, d_kps.ptr, big, d_desc.ptr, d_n.ptr, MB, MBF, d_u.ptr, d_d.ptr, sync=True)
    with pytest.raises(oa.OrbxError, match="capacity"):
        ex.stereo_match_batch_device_noalloc(2, d_kps.ptr, 16385, d_desc.ptr, d_n.ptr, MB, MBF, d_u.ptr, d_d.ptr, sync=True)
    ex.close()
    for a in (frames, d_kps, d_desc, d_n, d_u, d_d):
        a.free()
    # too many pairs for the array: 600 KITTI-shape pairs of kp_cap 2008 need 19 MiB
    w, h, n_pairs = 1241, 376, 600
    ex = oa.ORBextractor(2000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2 * n_pairs, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    n = 2 * n_pairs
    frames = DeviceArray((n, h, 1280), np.uint8, fill_byte=90)   # flat frames: no keypoints, the sizes are what matters
    d_kps, d_desc = DeviceArray((n, cap, 28), np.uint8), DeviceArray((n, cap, 32), np.uint8)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    d_u, d_d = DeviceArray((n_pairs, cap), np.float32), DeviceArray((n_pairs, cap), np.float32)
    ex.extract_batch_device(frames.ptr, n, w, h, 1280, h * 1280, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=True)
    with pytest.raises(oa.OrbxError, match="capacity"):
        ex.stereo_match_batch_device_noalloc(n_pairs, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, MB, MBF, d_u.ptr, d_d.ptr, sync=True)
    ex.stereo_match_batch_device_noalloc(200, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, MB, MBF, d_u.ptr, d_d.ptr, sync=True)
    ex.close()
    for a in (frames, d_kps, d_desc, d_n, d_u, d_d):
        a.free()


def _extract_pairs(ex, w, h, n_pairs, first_idx):
    """n_pairs fresh pairs extracted into device arrays; returns (host frames, device arrays...)."""
    from hip_rt import DeviceArray
    n = 2 * n_pairs
    stride = (w + 63) // 64 * 64
    host = np.zeros((n, h, stride), np.uint8)
    for p in range(n_pairs):
        l, r = _pair(w, h, 4, first_idx + p, 4 + 2 * p)
        host[2 * p, :, :w], host[2 * p + 1, :, :w] = l, r
    frames = DeviceArray.from_host(host)
    cap = ex.max_keypoints(w, h)
    d_kps, d_desc = DeviceArray((n, cap, 28), np.uint8), DeviceArray((n, cap, 32), np.uint8)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    d_u, d_d = DeviceArray((n_pairs, cap), np.float32, fill_byte=0x41), DeviceArray((n_pairs, cap), np.float32, fill_byte=0x41)
    ex.extract_batch_device(frames.ptr, n, w, h, stride, h * stride, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=True)
    return host, cap, frames, d_kps, d_desc, d_n, d_u, d_d


def test_noalloc_allocates_nothing_and_matches_the_oracle(oracle):
    from hip_rt import free_device_bytes
    w, h, nf, nl, n_pairs = 752, 480, 1200, 8, 3
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=2 * n_pairs, keep_pyramid=False)
    host, cap, frames, d_kps, d_desc, d_n, d_u, d_d = _extract_pairs(ex, w, h, n_pairs, 400)
    # this context has never run the allocating variant: its own stereo scratch does not exist, and must not after the call
    free_before = free_device_bytes()
    ex.stereo_match_batch_device_noalloc(n_pairs, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, MB, MBF, d_u.ptr, d_d.ptr, sync=True)
    assert free_device_bytes() == free_before, "the call changed the device's free memory"
    u, d, counts = d_u.to_host(), d_d.to_host(), d_n.to_host()
    for p in range(n_pairs):
        ol, orr = oracle.Oracle(nf, 1.2, nl, 20, 7), oracle.Oracle(nf, 1.2, nl, 20, 7)
        kl, _ = ol.extract(np.ascontiguousarray(host[2 * p, :, :w]))
        orr.extract(np.ascontiguousarray(host[2 * p + 1, :, :w]))
        ou, od, _, kept = oracle.stereo_match(ol, orr, np.float32(MB), np.float32(MBF))
        k = counts[2 * p]
        assert k == len(kl) and kept > 100
        assert (u[p, :k].view(np.uint32) == ou.view(np.uint32)).all(), "mvuRight, pair %d" % p
        assert (d[p, :k].view(np.uint32) == od.view(np.uint32)).all(), "mvDepth, pair %d" % p
    ex.close()
    for a in (frames, d_kps, d_desc, d_n, d_u, d_d):
        a.free()


def test_noalloc_two_contexts_share_the_array_in_turn():
    """Two contexts on one GPU, each calling from a thread of its own (the reference's two extractor instances,
    src/Frame.cc:78-81): the calls take turns on the scratch array, so each equals its allocating sibling."""
    import threading
    w, h, nf, nl, n_pairs = 752, 480, 1200, 8, 4
    ctxs = [oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=2 * n_pairs, keep_pyramid=False)
            for _ in range(2)]
    sets = [_extract_pairs(ex, w, h, n_pairs, 500 + 50 * i) for i, ex in enumerate(ctxs)]
    expect = []
    for ex, (_, cap, _, d_kps, d_desc, d_n, d_u, d_d) in zip(ctxs, sets):
        ex.stereo_match_batch_device(n_pairs, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, MB, MBF, d_u.ptr, d_d.ptr, sync=True)
        expect.append((d_u.to_host(), d_d.to_host()))
    errors = []

    def worker(i):
        try:
            ex, (_, cap, _, d_kps, d_desc, d_n, d_u, d_d) = ctxs[i], sets[i]
            for _ in range(25):
                ex.stereo_match_batch_device_noalloc(n_pairs, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, MB, MBF, d_u.ptr, d_d.ptr,
                                                     sync=False)
            ex.synchronize()
        except Exception as e:   # reported by the main thread
            errors.append(e)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for i, (_, cap, _, d_kps, d_desc, d_n, d_u, d_d) in enumerate(sets):
        counts = d_n.to_host()
        u, d = d_u.to_host(), d_d.to_host()
        for p in range(n_pairs):
            k = counts[2 * p]
            assert (u[p, :k].view(np.uint32) == expect[i][0][p, :k].view(np.uint32)).all(), "context %d pair %d" % (i, p)
            assert (d[p, :k].view(np.uint32) == expect[i][1][p, :k].view(np.uint32)).all(), "context %d pair %d" % (i, p)
    for ex in ctxs:
        ex.close()
    for s_ in sets:
        for a in s_[2:]:
            a.free()
