"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, stage by stage and end to end.

Integer stages (pyramid, blur, FAST candidates, quadtree retention, descriptors) must be
bit-exact; float outputs (angle, scaled coordinates) are produced by the same IEEE operation
sequence on both sides and are compared bit-exactly too (tolerance 0).
"""
import ctypes

import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth

pytestmark = pytest.mark.gpu

CONFIGS = [
    # (w, h, nfeatures, nlevels, config_id)
    (640, 480, 1000, 8, 1),
    (1241, 376, 2000, 8, 3),
    (752, 480, 1200, 8, 4),
    (317, 211, 500, 6, 7),
    (3840, 2160, 8000, 12, 5),   # 4K, 12 levels: largest quadtree tables and tile counts
]


def _check_stages(ex, orc, img, idx=0):
    nlev = ex.nlevels
    for l in range(nlev):
        got = ex.pyramid_level(idx, l)
        assert got.shape == orc.level(l).shape
        assert (got == orc.level(l)).all(), "pyramid level %d" % l
    for l in range(nlev):
        w, h = orc.level_dims(l)
        got = ex.debug_read(oa.DBG_BLURRED, idx, l).reshape(h, w)
        assert (got == orc.blurred(l)).all(), "blurred level %d" % l
    for l in range(nlev):
        got = ex.debug_read(oa.DBG_CANDIDATES, idx, l).view(np.int32).reshape(-1, 3)
        exp = orc.candidates(l)
        exp = np.stack([exp["x"], exp["y"], exp["response"]], 1)
        assert got.shape == exp.shape, "candidate count level %d: %s vs %s" % (l, got.shape, exp.shape)
        assert (got == exp).all(), "candidates level %d" % l


def _check_final(kps, desc, okps, odesc):
    assert len(kps) == len(okps)
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert (kps[f] == okps[f]).all(), f
    assert (desc == odesc).all()


@pytest.mark.parametrize("w,h,nf,nl,cid", CONFIGS)
def test_single_frame_all_stages(oracle, w, h, nf, nl, cid):
    img = synth.make_frame(w, h, cid, 0)
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=1)
    kps, desc = ex(img)
    orc = oracle.Oracle(nf, 1.2, nl, 20, 7)
    okps, odesc = orc.extract(img)
    _check_stages(ex, orc, img)
    for l in range(nl):
        got = ex.debug_read(oa.DBG_LEVEL_KPS, 0, l).view(np.int32).reshape(-1, 3)
        assert len(got) == orc.level_keypoint_count(l), "retained count level %d" % l
    _check_final(kps, desc, okps, odesc)
    # mvImagePyramid: level inside a 19-px reflect-101 border
    for l in range(nl):
        roi = ex.mvImagePyramid[l]
        padded = roi.base if roi.base is not None else roi
        assert (np.asarray(padded) == orc.padded_level(l)).all()
    ex.close()


def test_batch_matches_per_frame(oracle):
    w, h, nf, nl = 752, 480, 1200, 8
    n = 13      # not a multiple of 8: the last group of frames of the frame-major launches is partly empty
    imgs = synth.make_batch(n, w, h, 4, 10)
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=n)
    res = ex.extract_batch(imgs)
    orc = oracle.Oracle(nf, 1.2, nl, 20, 7)
    for i in range(n):
        okps, odesc = orc.extract(imgs[i])
        _check_final(res[i][0], res[i][1], okps, odesc)
    # intermediates of a frame in the middle of the batch
    orc.extract(imgs[3])
    _check_stages(ex, orc, imgs[3], idx=3)
    ex.close()


def test_pipelined_host_batch(oracle, monkeypatch):
    """Host batches of 16 frames or more are processed in chunks with uploads, extraction and downloads
    overlapped; here 40 frames in chunks of 7.  Results, pyramid read-back and the stereo stage must
    not depend on the chunking."""
    monkeypatch.setenv("ORBX_PIPE_CHUNK", "7")
    w, h, nf, nl, n = 640, 480, 1000, 8, 40
    base = synth.make_batch(5, w, h, 1, 40)
    imgs = [np.ascontiguousarray(base[i % 5] if i % 2 == 0 else np.roll(base[i % 5], -(3 + i % 7), axis=1)) for i in range(n)]
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=n)
    res = ex.extract_batch(imgs)
    orc = oracle.Oracle(nf, 1.2, nl, 20, 7)
    for i in range(n):                                 # every chunk, including the short last one
        okps, odesc = orc.extract(imgs[i])
        _check_final(res[i][0], res[i][1], okps, odesc)
    with pytest.raises(oa.OrbxError):                  # the candidate lists were reused chunk by chunk
        ex.debug_read(oa.DBG_CANDIDATES, 9, 0)
    orc.extract(imgs[9])                               # pyramid of a frame of the second chunk
    for l in range(nl):
        assert (ex.pyramid_level(9, l) == orc.level(l)).all()
    # stereo association between two frames of different chunks of the same context
    orr = oracle.Oracle(nf, 1.2, nl, 20, 7)
    orr.extract(imgs[33])                              # the scene of frame 8, shifted by 8 px
    orc.extract(imgs[8])
    u, d = ex.stereo_match(ex, 0.11, 47.9, img_left=8, img_right=33)
    ou, od, osad, _ = oracle.stereo_match(orc, orr, 0.11, 47.9)
    assert (osad >= 0).sum() > 50
    assert (u.view(np.uint32) == ou.view(np.uint32)).all() and (d.view(np.uint32) == od.view(np.uint32)).all()
    ex.close()


def test_pipelined_host_batch_two_threads_changing_sizes(oracle):
    """The host batch path runs a staging and a delivery thread beside the caller: two contexts called from two threads
    at the same time, with batch and chunk sizes that change from call to call, return for every frame what the
    oracle returns for it."""
    import ctypes as C
    import threading
    w, h, nf, nmax = 320, 240, 300, 40
    base = synth.make_batch(6, w, h, 2, 123)
    orc = oracle.Oracle(nf, 1.2, 6, 20, 7)
    expect = [orc.extract(f) for f in base]
    failures = []

    def worker(tid):
        rng = np.random.default_rng(tid)
        ex = oa.ORBextractor(nf, 1.2, 6, 20, 7, max_width=w, max_height=h, max_batch=nmax, keep_pyramid=False)
        cap = ex.max_keypoints(w, h)
        for call in range(12):
            n = int(rng.integers(16, nmax + 1))           # 16 frames and more take the pipelined path
            ex.set_debug_flag(oa.FLAG_PIPE_CHUNK, int(rng.choice([0, 5, 16, 33])))
            order = rng.integers(0, len(base), n)
            frames = [np.ascontiguousarray(base[i]) for i in order]
            kps, desc, cnt = np.zeros((n, cap), oa.KP_DTYPE), np.zeros((n, cap, 32), np.uint8), np.zeros(n, np.int32)
            ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
            strides = (C.c_int * n)(*[f.strides[0] for f in frames])
            rc = ex._L.orbx_extract_batch(ex._ctx, n, ptrs, w, h, strides, kps.ctypes.data_as(C.c_void_p), cap,
                                          desc.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p))
            if rc != 0:
                failures.append("thread %d call %d: rc %d" % (tid, call, rc))
                break
            for j, i in enumerate(order):
                okps, odesc = expect[int(i)]
                k = int(cnt[j])
                if k != len(okps) or kps[j, :k].tobytes() != okps.tobytes() or not (desc[j, :k] == odesc).all():
                    failures.append("thread %d call %d frame %d" % (tid, call, j))
        ex.close()

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures, failures[:5]


def test_strided_input_and_reuse(oracle):
    w, h = 640, 480
    big = np.zeros((h, w + 37), np.uint8)
    big[:, :w] = synth.make_frame(w, h, 1, 5)
    view = big[:, :w]
    assert not view.flags["C_CONTIGUOUS"]
    ex = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_width=1241, max_height=480, max_batch=2)
    orc = oracle.Oracle(1000, 1.2, 8, 20, 7)
    kps, desc = ex(view)
    _check_final(kps, desc, *orc.extract(view))
    # same context, another size, then back
    img2 = synth.make_frame(1241, 376, 3, 2)
    _check_final(*ex(img2), *orc.extract(img2))
    _check_final(*ex(view), *orc.extract(view))
    ex.close()


def test_flat_and_sparse_images(oracle):
    w, h = 320, 240
    ex = oa.ORBextractor(500, 1.2, 4, 20, 7, max_width=w, max_height=h)
    orc = oracle.Oracle(500, 1.2, 4, 20, 7)
    flat = np.full((h, w), 90, np.uint8)
    kps, desc = ex(flat)
    assert len(kps) == 0 and desc.shape == (0, 32)
    assert len(orc.extract(flat)[0]) == 0
    # a handful of isolated blobs: fewer candidates than the quota, min-threshold cells
    sparse = flat.copy()
    rng = np.random.default_rng(3)
    for _ in range(25):
        x, y = rng.integers(30, w - 30), rng.integers(30, h - 30)
        sparse[y:y + 3, x:x + 3] = 90 + rng.integers(9, 120)
    _check_final(*ex(sparse), *orc.extract(sparse))
    assert ex(np.zeros((0, 0), np.uint8)) == (None, None)
    ex.close()


def test_dense_noise_images(oracle):
    """Images in which most pixels pass the rejection test: the survivor queue of a tile overflows and
    the kernel's every-pixel path runs; salt-and-pepper noise puts many pixels on both polarities."""
    w, h = 400, 300
    rng = np.random.default_rng(11)
    white = rng.integers(0, 256, (h, w), dtype=np.uint8)
    salt = np.full((h, w), 128, np.uint8)
    m = rng.random((h, w))
    salt[m < 0.25] = 0
    salt[m > 0.75] = 255
    checker = (((np.add.outer(np.arange(h), np.arange(w)) & 1) * 200) + 20).astype(np.uint8)
    half = white.copy()
    half[:, w // 2:] = synth.make_frame(w, h, 2, 1)[:, w // 2:]
    ex = oa.ORBextractor(800, 1.2, 5, 20, 7, max_width=w, max_height=h)
    orc = oracle.Oracle(800, 1.2, 5, 20, 7)
    for img in (white, salt, checker, half):
        kps, desc = ex(img)
        okps, odesc = orc.extract(img)
        _check_stages(ex, orc, img)
        _check_final(kps, desc, okps, odesc)
    ex.close()


def test_full_size_batch_properties(oracle):
    """The benchmark configuration itself: 512 KITTI-shape frames in one batch through the
    device-pointer entry point.  Size-independent properties: a second run reproduces every output
    byte; frames that are copies of each other give identical results wherever they sit in the
    batch; sampled frames equal the oracle bit for bit; counts respect nfeatures and the capacity."""
    from hip_rt import DeviceArray
    w, h, nf, nl, n = 1241, 376, 2000, 8, 512
    stride = 1280
    base = synth.make_batch(16, w, h, 3, 100)
    host = np.zeros((n, h, stride), np.uint8)
    for i in range(n):
        host[i, :, :w] = base[i % 16] if i < 480 else 255 - base[i % 16]
    frames = DeviceArray.from_host(host)
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    outs = []
    for rep in range(2):
        d_kps = DeviceArray((n, cap, 28), np.uint8, fill_byte=0)
        d_desc = DeviceArray((n, cap, 32), np.uint8, fill_byte=0)
        d_n = DeviceArray((n,), np.int32, fill_byte=0)
        ex.extract_batch_device(frames.ptr, n, w, h, stride, h * stride, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=True)
        outs.append((d_kps.to_host(), d_desc.to_host(), d_n.to_host()))
        for a in (d_kps, d_desc, d_n):
            a.free()
    (k0, q0, c0), (k1, q1, c1) = outs
    assert (c0 == c1).all() and (c0 >= nf).all() and (c0 <= cap).all()
    valid = np.arange(cap)[None, :] < c0[:, None]
    assert (k0[valid] == k1[valid]).all() and (q0[valid] == q1[valid]).all()        # reproducible
    for i in (16, 33, 250, 479):                                                    # copies agree
        j = i % 16
        assert c0[i] == c0[j]
        assert (k0[i, :c0[i]] == k0[j, :c0[j]]).all() and (q0[i, :c0[i]] == q0[j, :c0[j]]).all()
    orc = oracle.Oracle(nf, 1.2, nl, 20, 7)
    for i in (0, 7, 481, 511):                                                      # oracle parity
        okps, odesc = orc.extract(host[i, :, :w])
        assert c0[i] == len(okps)
        _check_final(np.ascontiguousarray(k0[i, :c0[i]]).reshape(-1).view(oa.KP_DTYPE), q0[i, :c0[i]], okps, odesc)
    ex.close()
    frames.free()


def test_back_to_back_async_batches(oracle):
    """Several batches enqueued without waiting in between (sync=False): the library overlaps a batch's
    descriptor stage with the next batch's detection on another stream, and its arenas are reused by
    every batch, so this is the ordering test of that schedule.  Inputs differ per batch, outputs go to
    separate buffers, everything is compared with the oracle after one final synchronisation."""
    from hip_rt import DeviceArray
    w, h, nf, nl, n, nb = 752, 480, 1200, 8, 24, 5
    stride = 768
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    hosts, d_in, d_out = [], [], []
    for b in range(nb):
        host = np.zeros((n, h, stride), np.uint8)
        host[:, :, :w] = synth.make_batch(n, w, h, 4, 200 + 37 * b)
        hosts.append(host)
        d_in.append(DeviceArray.from_host(host))
        d_out.append((DeviceArray((n, cap, 28), np.uint8, fill_byte=0), DeviceArray((n, cap, 32), np.uint8, fill_byte=0),
                      DeviceArray((n,), np.int32, fill_byte=0)))
    for rep in range(2):                                  # twice: the second round starts on warm streams
        for b in range(nb):
            k, q, c = d_out[b]
            ex.extract_batch_device(d_in[b].ptr, n, w, h, stride, h * stride, k.ptr, cap, q.ptr, c.ptr, sync=False)
    ex.synchronize()
    orc = oracle.Oracle(nf, 1.2, nl, 20, 7)
    for b in range(nb):
        kps, desc, cnt = (a.to_host() for a in d_out[b])
        for i in (0, 11, n - 1):
            okps, odesc = orc.extract(hosts[b][i, :, :w])
            assert cnt[i] == len(okps), (b, i)
            _check_final(np.ascontiguousarray(kps[i, :cnt[i]]).reshape(-1).view(oa.KP_DTYPE), desc[i, :cnt[i]], okps, odesc)
    ex.close()
    for a in d_in:
        a.free()
    for t in d_out:
        for a in t:
            a.free()


def test_free_running_batches_mixed_with_staging_calls(oracle):
    """Batches whose images are resident when the call is made build their pyramid ahead of the main stream;
    a batch whose images a staging call on the same context has just produced (orbx_gray_batch_device) must
    not, and neither must the batches around a call that reads the arenas from the main stream (stereo
    association of the previous batch).  All of them are enqueued without waiting; the results must equal
    those of the same calls made one at a time, and those of the ordered schedule (ORBX_FLAG_ORDERED)."""
    from hip_rt import DeviceArray
    w, h, nf, nl, n = 640, 480, 1000, 8, 8
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=n, keep_pyramid=False)
    cap = ex.max_keypoints(w, h)
    gray = [synth.make_batch(n, w, h, 4, 900 + 13 * b) for b in range(4)]
    rgb = np.repeat(gray[2][..., None], 3, axis=3)          # R = G = B: the conversion returns the grey value
    d_gray = [DeviceArray.from_host(g) for g in gray]
    d_rgb = DeviceArray.from_host(rgb)
    d_conv = DeviceArray((n, h, w), np.uint8, fill_byte=0)
    d_u = DeviceArray((n // 2, cap), np.float32, fill_byte=0)
    d_z = DeviceArray((n // 2, cap), np.float32, fill_byte=0)

    def run(wait):
        outs = [(DeviceArray((n, cap, 28), np.uint8, fill_byte=0), DeviceArray((n, cap, 32), np.uint8, fill_byte=0),
                 DeviceArray((n,), np.int32, fill_byte=0)) for _ in range(6)]
        def batch(i, ptr):
            k, q, c = outs[i]
            ex.extract_batch_device(ptr, n, w, h, w, h * w, k.ptr, cap, q.ptr, c.ptr, sync=wait)
        d_conv.fill(0)
        batch(0, d_gray[0].ptr)
        batch(1, d_gray[1].ptr)
        ex.gray_batch_device(n, d_rgb.ptr, w, h, 3 * w, 3 * w * h, 3, 0, d_conv.ptr, w, w * h, sync=wait)
        batch(2, d_conv.ptr)                                 # its images come from the kernel in front of it
        batch(3, d_gray[3].ptr)
        k, q, c = outs[3]
        ex.stereo_match_batch_device(n // 2, k.ptr, cap, q.ptr, c.ptr, 0.5, 400.0, d_u.ptr, d_z.ptr, sync=wait)
        batch(4, d_gray[0].ptr)                              # the next two reuse the arenas the association reads
        batch(5, d_gray[1].ptr)
        ex.synchronize()
        res = [tuple(a.to_host() for a in t) for t in outs] + [(d_u.to_host(), d_z.to_host())]
        for t in outs:
            for a in t:
                a.free()
        return res

    ref = run(True)
    for ordered in (0, 1, 0):
        ex.set_debug_flag(oa.FLAG_ORDERED, ordered)
        got = run(False)
        for r, g in zip(ref, got):
            for a, b in zip(r, g):
                assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), ordered
    orc = oracle.Oracle(nf, 1.2, nl, 20, 7)
    kps, desc, cnt = ref[2]
    okps, odesc = orc.extract(gray[2][3])
    assert cnt[3] == len(okps)
    _check_final(np.ascontiguousarray(kps[3, :cnt[3]]).reshape(-1).view(oa.KP_DTYPE), desc[3, :cnt[3]], okps, odesc)
    ex.close()
    for a in d_gray + [d_rgb, d_conv, d_u, d_z]:
        a.free()


def test_dotted_images(oracle):
    """Single-pixel dots on a flat ground: few pixels pass the rejection test but most of those are
    corners, so a tile's corner list runs from nearly empty to over its capacity (step 3 of the FAST
    kernel then walks the whole queue); stripes of low-contrast dots are only found by the retry at
    the minimum threshold, in tiles that already hold survivors of the first pass."""
    w, h = 420, 300
    ex = oa.ORBextractor(3000, 1.2, 4, 20, 7, max_width=w, max_height=h)
    orc = oracle.Oracle(3000, 1.2, 4, 20, 7)
    rng = np.random.default_rng(29)
    low = (np.arange(w) // 45) % 2 == 1                  # every other 45-pixel stripe: contrast 12
    for density in (0.02, 0.05, 0.08, 0.12, 0.2):
        img = np.full((h, w), 100, np.uint8)
        dots = rng.random((h, w)) < density
        img[dots] = 200
        img[dots & low[None, :]] = 112
        kps, desc = ex(img)
        okps, odesc = orc.extract(img)
        _check_stages(ex, orc, img)
        _check_final(kps, desc, okps, odesc)
        assert len(kps) > 500
    ex.close()


@pytest.mark.parametrize("w,h,nf,scale,nl,ini,mn,cid", [
    (200, 150, 50, 1.2, 1, 20, 7, 11),        # one level, tiny quota
    (255, 193, 300, 1.5, 3, 20, 7, 12),       # coarse scale factor, odd sizes
    (512, 512, 700, 2.0, 4, 20, 7, 13),       # scale 2, square image (one quadtree root)
    (1000, 130, 400, 1.2, 4, 20, 7, 14),      # very wide strip: seven or eight quadtree roots
    (640, 480, 1000, 1.2, 8, 12, 12, 15),     # equal thresholds: the retry finds nothing new
    (640, 480, 1000, 1.2, 8, 60, 30, 16),     # high thresholds: many cells stay empty, quotas not reached
    (640, 480, 5000, 1.1, 12, 20, 7, 17),     # fine scale factor, 12 levels, quota above the candidate count
    (131, 119, 100, 1.2, 4, 20, 7, 18),       # top level just above the 62-px limit of the cell grid
])
def test_parameter_space(oracle, w, h, nf, scale, nl, ini, mn, cid):
    img = synth.make_frame(w, h, cid, 1)
    ex = oa.ORBextractor(nf, scale, nl, ini, mn, max_width=w, max_height=h)
    orc = oracle.Oracle(nf, scale, nl, ini, mn)
    kps, desc = ex(img)
    okps, odesc = orc.extract(img)
    _check_stages(ex, orc, img)
    _check_final(kps, desc, okps, odesc)
    assert len(kps) > 0
    ex.close()


@pytest.mark.parametrize("w,h,nf,nl,cid", [
    (640, 480, 3000, 1, 21),       # one level that keeps 3000 features
    (1168, 665, 6000, 2, 22),      # 3273 + 2727
    (1500, 1100, 8000, 1, 23),     # the largest quota a level can have is 8189
])
def test_quotas_beyond_the_lds_tables(oracle, w, h, nf, nl, cid):
    """A level whose quota is above about 2500 features does not fit the quadtree's LDS tables: the kernel then keeps
    them in device memory (same code, global atomics).  Single frame and a batch of three."""
    imgs = [synth.make_frame(w, h, cid, k) for k in range(3)]
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=3)
    orc = oracle.Oracle(nf, 1.2, nl, 20, 7)
    kps, desc = ex(imgs[0])
    okps, odesc = orc.extract(imgs[0])
    _check_stages(ex, orc, imgs[0])
    _check_final(kps, desc, okps, odesc)
    assert len(kps) > 0.7 * nf
    res = ex.extract_batch(imgs)
    for k in range(3):
        okps, odesc = orc.extract(imgs[k])
        _check_final(res[k][0], res[k][1], okps, odesc)
    ex.close()


def test_resize_tiled_kernel(oracle, monkeypatch):
    """The pyramid has three resize kernels: the streaming one (taps of four columns within eight source
    bytes), the LDS-tiled one for everything else, and for a launch of one small frame the one that produces
    three levels per launch (the single-frame tests above run through it, the batch tests through the first
    two).  The diagnostic switch forces the tiled one."""
    monkeypatch.setenv("ORBX_RESIZE_TILES", "1")
    w, h = 752, 480
    img = synth.make_frame(w, h, 4, 3)
    ex = oa.ORBextractor(1200, 1.2, 8, 20, 7, max_width=w, max_height=h)
    orc = oracle.Oracle(1200, 1.2, 8, 20, 7)
    kps, desc = ex(img)
    okps, odesc = orc.extract(img)
    _check_stages(ex, orc, img)
    _check_final(kps, desc, okps, odesc)
    ex.close()


def test_device_input_unaligned(oracle):
    """Device-resident input whose pointer, stride and frame stride are not multiples of 4: the library
    cannot use it in place and copies it into its arena first."""
    from hip_rt import DeviceArray
    w, h, nf, nl, n = 641, 479, 1000, 8, 3
    stride, fstride, lead = 645, 645 * 479 + 7, 3
    imgs = synth.make_batch(n, w, h, 1, 300)
    raw = np.zeros(lead + n * fstride, np.uint8)
    for i in range(n):
        view = raw[lead + i * fstride: lead + i * fstride + h * stride].reshape(h, stride)
        view[:, :w] = imgs[i]
    d_raw = DeviceArray.from_host(raw)
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=n)
    cap = ex.max_keypoints(w, h)
    d_kps, d_desc = DeviceArray((n, cap, 28), np.uint8, fill_byte=0), DeviceArray((n, cap, 32), np.uint8, fill_byte=0)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    for rep in range(2):
        ex.extract_batch_device(d_raw.ptr + lead, n, w, h, stride, fstride, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=True)
    cnt, kps, desc = d_n.to_host(), d_kps.to_host(), d_desc.to_host()
    orc = oracle.Oracle(nf, 1.2, nl, 20, 7)
    for i in range(n):
        okps, odesc = orc.extract(imgs[i])
        assert cnt[i] == len(okps)
        _check_final(np.ascontiguousarray(kps[i, :cnt[i]]).reshape(-1).view(oa.KP_DTYPE), desc[i, :cnt[i]], okps, odesc)
    assert (ex.pyramid_level(2, 0) == imgs[2]).all()
    ex.close()
    for a in (d_raw, d_kps, d_desc, d_n):
        a.free()


def test_quadtree_nodes_above_65535_points(oracle):
    """A 2160 x 2160 frame of single-pixel dots four pixels apart: every dot is a FAST corner that survives
    the suppression, so level 0 has a quarter of a million candidates under ONE quadtree root.  The first
    sweep leaves four nodes; with a level-0 quota of 5 the one-at-a-time phase then splits exactly one of
    them, the largest.  Rows of dots are removed so that the upper left node (70225 points) is the largest
    but, modulo 65536, the smallest: the order must come from the real sizes."""
    w, h, nf, nl = 2160, 2160, 25, 12
    img = np.full((h, w), 100, np.uint8)
    img[2::4, 2::4] = 200
    img[2:2 + 4 * 24:4, w // 2:] = 100                   # upper right: 24 rows of dots less
    img[h // 2 + 2:h // 2 + 2 + 4 * 44:4, :] = 100       # lower half: 44 rows less
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=1, keep_pyramid=False)
    orc = oracle.Oracle(nf, 1.2, nl, 20, 7)
    kps, desc = ex(img)
    okps, odesc = orc.extract(img)
    assert ex.mnFeaturesPerLevel[0] == 5
    cand = orc.candidates(0)
    mx, my = (w - 32 + 1) // 2, (h - 32 + 1) // 2
    size = [int((((cand["x"] < mx) == a) & ((cand["y"] < my) == b)).sum()) for a, b in
            ((True, True), (False, True), (True, False), (False, False))]
    assert size[0] > 65535 > size[1] > size[2] and (size[0] & 0xFFFF) < size[3], size
    got = ex.debug_read(oa.DBG_LEVEL_KPS, 0, 0).view(np.int32).reshape(-1, 3)
    assert len(got) == orc.level_keypoint_count(0) == 7
    assert ((got[:, 0] < mx + 16) & (got[:, 1] < my + 16)).sum() == 4   # the upper left node was the one split
    _check_final(kps, desc, okps, odesc)
    ex.close()


def test_single_frame_launch_graph_and_plain_launches(oracle):
    """orbx_extract replays a captured HIP graph per image size; the no-graph switch issues the same launches
    one by one.  Both paths, sizes changing back and forth on one context, an asynchronous device batch in
    between (its arenas are the graph's), then the graph again: all equal to the oracle."""
    from hip_rt import DeviceArray
    sizes = [(640, 480, 1), (752, 480, 4), (640, 480, 1)]
    ex = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_width=752, max_height=480, max_batch=4)
    orc = oracle.Oracle(1000, 1.2, 8, 20, 7)
    for use_graph in (True, False, True):
        ex.set_debug_flag(oa.FLAG_NO_GRAPH, 0 if use_graph else 1)
        for rep, (w, h, cid) in enumerate(sizes):
            for k in range(3):                           # the same size several times: capture once, replay
                img = synth.make_frame(w, h, cid, 40 + 3 * rep + k)
                kps, desc = ex(img)
                okps, odesc = orc.extract(img)
                _check_final(kps, desc, okps, odesc)
            _check_stages(ex, orc, img)
    # an asynchronous batch through the device entry point, then the graph on top of it at once
    w, h, n = 752, 480, 4
    host = np.zeros((n, h, 768), np.uint8)
    host[:, :, :w] = synth.make_batch(n, w, h, 4, 70)
    d_in = DeviceArray.from_host(host)
    cap = ex.max_keypoints(w, h)
    d_kps, d_desc = DeviceArray((n, cap, 28), np.uint8, fill_byte=0), DeviceArray((n, cap, 32), np.uint8, fill_byte=0)
    d_n = DeviceArray((n,), np.int32, fill_byte=0)
    img = synth.make_frame(w, h, 4, 90)
    ex(img)                                              # graph captured for this size
    ex.extract_batch_device(d_in.ptr, n, w, h, 768, h * 768, d_kps.ptr, cap, d_desc.ptr, d_n.ptr, sync=False)
    kps, desc = ex(img)
    _check_final(kps, desc, *orc.extract(img))
    cnt, bk, bd = d_n.to_host(), d_kps.to_host(), d_desc.to_host()
    for i in range(n):
        okps, odesc = orc.extract(host[i, :, :w])
        assert cnt[i] == len(okps)
        _check_final(np.ascontiguousarray(bk[i, :cnt[i]]).reshape(-1).view(oa.KP_DTYPE), bd[i, :cnt[i]], okps, odesc)
    ex.close()
    for a in (d_in, d_kps, d_desc, d_n):
        a.free()


def test_extract_pair_one_graph(oracle):
    """orbx_extract_pair: both images of a pair and the stereo association as one call on one context, against the
    oracle; a change of mb / mbf re-captures; plain launches (the no-graph switch) give the same."""
    w, h, nf, nl = 1241, 376, 2000, 8
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=2, keep_pyramid=False)
    ol, orr = oracle.Oracle(nf, 1.2, nl, 20, 7), oracle.Oracle(nf, 1.2, nl, 20, 7)
    for i, (mb, mbf, plain) in enumerate(((0.5371657, 386.1448, 0), (0.5371657, 386.1448, 0), (0.3, 200.0, 0),
                                          (0.5371657, 386.1448, 1))):
        wide = synth.make_frame(w + 32, h, 3, 60 + i)
        left = np.ascontiguousarray(wide[:, :w])
        right = np.ascontiguousarray(np.clip(wide[:, 7:w + 7].astype(np.int32) +
                                             np.random.default_rng(i).integers(-2, 3, (h, w)), 0, 255).astype(np.uint8))
        ex.set_debug_flag(oa.FLAG_NO_GRAPH, plain)
        (kl, dl), (kr, dr), u, d = ex.extract_pair(left, right, mb, mbf)
        okl, odl = ol.extract(left)
        okr, odr = orr.extract(right)
        _check_final(kl, dl, okl, odl)
        _check_final(kr, dr, okr, odr)
        ou, od, _, kept = oracle.stereo_match(ol, orr, np.float32(mb), np.float32(mbf))
        assert kept > 200
        assert (u.view(np.uint32) == ou.view(np.uint32)).all() and (d.view(np.uint32) == od.view(np.uint32)).all()
        # the pyramids of both images are there for read-backs
        assert (ex.pyramid_level(1, 3) == orr.level(3)).all() and (ex.pyramid_level(0, 7) == ol.level(7)).all()
    # without the association
    (kl, dl), (kr, dr), u, d = ex.extract_pair(left, right)
    assert u is None and d is None
    _check_final(kl, dl, okl, odl)
    _check_final(kr, dr, okr, odr)
    ex.close()
    one = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=1, keep_pyramid=False)
    with pytest.raises(oa.OrbxError, match="capacity"):
        one.extract_pair(left, right)
    one.close()


def test_padded_pyramid_built_on_the_device(oracle):
    """mvImagePyramid's layout (19-px reflect-101 rim around every level) comes out of the extraction's own submission,
    in the graph path and with plain launches, and follows a change of image size."""
    ex = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_width=752, max_height=480, max_batch=1, keep_pyramid=True)
    orc = oracle.Oracle(1000, 1.2, 8, 20, 7)
    for (w, h, idx, plain) in ((640, 480, 1, 0), (640, 480, 2, 0), (752, 480, 3, 0), (400, 300, 4, 1), (640, 480, 5, 1)):
        ex.set_debug_flag(oa.FLAG_NO_GRAPH, plain)
        img = synth.make_frame(w, h, 1, idx)
        kps, desc = ex(img)
        okps, odesc = orc.extract(img)
        _check_final(kps, desc, okps, odesc)
        for l in range(8):
            view = ex.padded_level(l, copy=False)
            lw, lh = orc.level_dims(l)
            assert view.shape == (lh, lw) and view.strides[0] == lw + 38
            base = np.frombuffer((ctypes.c_uint8 * ((lw + 38) * (lh + 38))).from_address(view.ctypes.data - 19 * (lw + 38) - 19),
                                 np.uint8).reshape(lh + 38, lw + 38)
            assert (base == orc.padded_level(l)).all(), "padded level %d of %dx%d" % (l, w, h)
            assert (ex.mvImagePyramid[l] == orc.level(l)).all()
    ex.close()


def test_errors():
    ex = oa.ORBextractor(500, 1.2, 8, 20, 7, max_width=320, max_height=240)
    with pytest.raises(oa.OrbxError):
        ex(np.zeros((480, 640), np.uint8))          # larger than the context
    with pytest.raises(oa.OrbxError):
        ex(np.zeros((240, 320, 3), np.uint8))       # not single channel
    with pytest.raises(oa.OrbxError):
        ex(np.zeros((100, 100), np.uint8))          # top level too small for a 30-px cell grid
    ex.close()
