"""GPU parity of the stereo association stage (Frame::ComputeStereoMatches, reference
src/Frame.cc:485-659) against the CPU oracle: mvuRight and mvDepth bit-exact (tolerance 0: both
sides evaluate the same IEEE float32 statements on identical integer inputs)."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth

pytestmark = pytest.mark.gpu

MB, MBF = 0.5371657, 386.1448  # KITTI 00-02: bf = 386.1448, fx = 718.856


def shifted_pair(w, h, cid, idx, disparity_px):
    """Left = synthetic frame, right = the same scene seen with a depth-dependent shift: the top
    half moves by disparity_px, the bottom half by twice that, plus mild noise."""
    left = synth.make_frame(w + 64, h, cid, idx)
    rng = np.random.default_rng(idx)
    right = np.empty((h, w), np.uint8)
    half = h // 2
    right[:half] = left[:half, disparity_px:disparity_px + w]
    right[half:] = left[half:, 2 * disparity_px:2 * disparity_px + w]
    noise = rng.integers(-3, 4, (h, w))
    right = np.clip(right.astype(np.int32) + noise, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(left[:, :w]), right


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("w,h,nf,nl,disp", [(1241, 376, 2000, 8, 9), (752, 480, 1200, 8, 5), (640, 480, 1000, 8, 0)])
def test_stereo_two_contexts(oracle, w, h, nf, nl, disp):
    left, right = shifted_pair(w, h, 3, 7, disp)
    exl = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h)
    exr = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h)
    exl(left)
    exr(right)
    u, d = exl.stereo_match(exr, MB, MBF)
    ol, orr = oracle.Oracle(nf, 1.2, nl, 20, 7), oracle.Oracle(nf, 1.2, nl, 20, 7)
    kl, _ = ol.extract(left)
    orr.extract(right)
    ou, od, osad, kept = oracle.stereo_match(ol, orr, MB, MBF)
    assert len(u) == len(kl)
    assert kept > 100, kept                      # the case exercises real matches
    assert ((ou >= 0).sum() == kept)
    assert (bits(u) == bits(ou)).all()
    assert (bits(d) == bits(od)).all()
    exl.close()
    exr.close()


def test_stereo_border_loader(oracle, monkeypatch):
    """Windows that reach into a level's border are loaded byte by byte with reflection; the
    diagnostic switch sends every window that way, and the results must not change."""
    monkeypatch.setenv("ORBX_STEREO_BORDER_PATH", "1")
    test_stereo_two_contexts(oracle, 752, 480, 1200, 8, 5)


def test_stereo_batch_device(oracle):
    from hip_rt import DeviceArray
    w, h, nf, nl = 752, 480, 1200, 8
    n_pairs = 3
    imgs = []
    for p in range(n_pairs):
        l, r = shifted_pair(w, h, 4, 20 + p, 4 + 3 * p)
        imgs += [l, r]
    n = 2 * n_pairs
    stride = (w + 63) // 64 * 64
    host = np.zeros((n, h, stride), np.uint8)
    host[:, :, :w] = np.stack(imgs)
    frames = DeviceArray.from_host(host)
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    d_kps = DeviceArray((n, cap, 28), np.uint8)
    d_desc = DeviceArray((n, cap, 32), np.uint8)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    d_u = DeviceArray((n_pairs, cap), np.float32, fill_byte=0x41)
    d_d = DeviceArray((n_pairs, cap), np.float32, fill_byte=0x41)
    ex.extract_batch_device(frames.ptr, n, w, h, stride, h * stride, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=False)
    ex.stereo_match_batch_device(n_pairs, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, MB, MBF, d_u.ptr, d_d.ptr, sync=True)
    counts = d_n.to_host()
    u_all, d_all = d_u.to_host(), d_d.to_host()
    for p in range(n_pairs):
        ol, orr = oracle.Oracle(nf, 1.2, nl, 20, 7), oracle.Oracle(nf, 1.2, nl, 20, 7)
        kl, _ = ol.extract(imgs[2 * p])
        orr.extract(imgs[2 * p + 1])
        ou, od, _, kept = oracle.stereo_match(ol, orr, MB, MBF)
        k = counts[2 * p]
        assert k == len(kl) and kept > 50
        assert (bits(u_all[p, :k]) == bits(ou)).all(), "pair %d" % p
        assert (bits(d_all[p, :k]) == bits(od)).all(), "pair %d" % p
    ex.close()
    for a in (frames, d_kps, d_desc, d_n, d_u, d_d):
        a.free()


def test_stereo_no_matches_and_errors(oracle):
    w, h = 640, 480
    a = synth.make_frame(w, h, 1, 0)
    b = synth.make_frame(w, h, 1, 99)            # unrelated scene: few or no accepted matches
    exl = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h)
    exr = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h)
    with pytest.raises(oa.OrbxError):
        exl.stereo_match(exr, MB, MBF)           # nothing extracted yet
    exl(a)
    exr(b)
    u, d = exl.stereo_match(exr, MB, MBF)
    ol, orr = oracle.Oracle(1000, 1.2, 8, 20, 7), oracle.Oracle(1000, 1.2, 8, 20, 7)
    ol.extract(a)
    orr.extract(b)
    ou, od, _, _ = oracle.stereo_match(ol, orr, MB, MBF)
    assert (bits(u) == bits(ou)).all() and (bits(d) == bits(od)).all()
    # a flat right image has no keypoints at all
    exr(np.full((h, w), 80, np.uint8))
    u, d = exl.stereo_match(exr, MB, MBF)
    assert (u == -1).all() and (d == -1).all()
    with pytest.raises(oa.OrbxError):
        exl.stereo_match(exr, 0.0, MBF)          # baseline must be positive
    exl.close()
    exr.close()
