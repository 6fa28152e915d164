"""Known-answer tests of the CPU restatement of the stereo association stage
(Frame::ComputeStereoMatches, reference src/Frame.cc:485-659; ORBmatcher::DescriptorDistance,
src/ORBmatcher.cc:1684-1700).  Expectations are derived by hand from the reference's statements,
not from the restatement."""
import numpy as np

from orb_slam2_amd import synth

MB, MBF = 0.5371657, 386.1448


def test_descriptor_distance_is_popcount(oracle):
    rng = np.random.default_rng(5)
    z = np.zeros(32, np.uint8)
    assert oracle.descriptor_distance(z, z) == 0
    assert oracle.descriptor_distance(z, np.full(32, 255, np.uint8)) == 256
    one = z.copy()
    one[17] = 0x10
    assert oracle.descriptor_distance(z, one) == 1
    for _ in range(50):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert oracle.descriptor_distance(a, b) == int(np.unpackbits(a ^ b).sum())


def test_identical_images_are_all_cut_by_a_zero_median(oracle):
    """Left == right: every keypoint's best candidate is itself (Hamming 0) and the window distance
    is 0 at shift 0, so every accepted match carries distance 0, the median is 0, the threshold
    1.5f*1.4f*0 is 0 and `dist < 0` holds for none: the cut loop (:649-657) resets them all."""
    img = synth.make_frame(640, 480, 1, 3)
    a, b = oracle.Oracle(1000, 1.2, 8), oracle.Oracle(1000, 1.2, 8)
    kps, _ = a.extract(img)
    b.extract(img)
    u, d, sad, kept = oracle.stereo_match(a, b, MB, MBF)
    assert (sad >= 0).sum() > 0.2 * len(kps)      # matches were found and accepted ...
    assert (sad[sad >= 0] == 0).all()             # ... all with a zero window distance ...
    assert kept == 0 and (u == -1).all() and (d == -1).all()   # ... and all cut


def test_shift_is_recovered(oracle):
    """Right image = left scene shifted by s px plus mild noise: matched level-0 keypoints have a
    disparity within a pixel of s (the parabola moves it by less than one shift step), and every
    output obeys depth = mbf / (uL - uR) in float32 (:629,638)."""
    s = 8
    wide = synth.make_frame(640 + 32, 480, 1, 9)
    left = np.ascontiguousarray(wide[:, :640])
    right = wide[:, s:640 + s].astype(np.int32) + np.random.default_rng(2).integers(-4, 5, (480, 640))
    right = np.clip(right, 0, 255).astype(np.uint8)
    a, b = oracle.Oracle(1000, 1.2, 8), oracle.Oracle(1000, 1.2, 8)
    kps, _ = a.extract(left)
    b.extract(right)
    u, d, sad, kept = oracle.stereo_match(a, b, MB, MBF)
    m = u >= 0
    assert kept == m.sum() and kept > 100
    disp = kps["x"][m] - u[m]
    lvl0 = kps["octave"][m] == 0
    assert lvl0.sum() > 20 and (np.abs(disp[lvl0] - s) <= 1.0).all()
    assert (np.abs(disp - s) <= 1.0 * 1.2 ** kps["octave"][m] + 1e-3).all()
    assert (d[m] == np.float32(MBF) / disp.astype(np.float32)).all()
    assert (disp > 0).all() and (disp < np.float32(MBF) / np.float32(MB)).all()


def test_median_cut_and_empty_inputs(oracle):
    """No right keypoints -> all -1 (the reference would index an empty vector at :646; pinned as a
    no-op).  With matches, everything at or above 1.5f*1.4f*median of the accepted window
    distances is reset to -1 (:645-658)."""
    img = synth.make_frame(640, 480, 1, 4)
    flat = np.full((480, 640), 77, np.uint8)
    a, b = oracle.Oracle(1000, 1.2, 8), oracle.Oracle(1000, 1.2, 8)
    a.extract(img)
    b.extract(flat)
    u, d, sad, kept = oracle.stereo_match(a, b, MB, MBF)
    assert kept == 0 and (u == -1).all() and (d == -1).all() and (sad == -1).all()
    noisy = np.clip(img.astype(np.int32) + np.random.default_rng(1).integers(-6, 7, img.shape), 0, 255).astype(np.uint8)
    b.extract(noisy)
    u, d, sad, kept = oracle.stereo_match(a, b, MB, MBF)
    acc = np.sort(sad[sad >= 0])
    assert len(acc) > 50
    th = np.float32(np.float32(1.5) * np.float32(1.4)) * np.float32(acc[len(acc) // 2])
    cut = (sad >= 0) & ~(sad.astype(np.float32) < th)
    assert (u[cut] == -1).all() and (d[cut] == -1).all()
    keep = (sad >= 0) & ~cut
    assert (u[keep] >= 0).all() and kept == keep.sum()
