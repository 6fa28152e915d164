"""CPU-only checks of the C ABI library: it loads, exports every declared symbol, and its host
logic (constructor tables, level geometry, resize taps) agrees with the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import orb_slam2_amd as oa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = oa.load_library()
    hdr = open(os.path.join(ROOT, "include", "orbx.h")).read()
    declared = sorted(set(re.findall(r"\b(orbx_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), name
    assert set(declared) == set(oa.EXPORTS)
    assert lib.orbx_version().startswith(b"orbx")
    assert lib.orbx_strerror(0) == b"ok" and lib.orbx_strerror(-4) == b"capacity exceeded"


@pytest.mark.parametrize("nf,nl", [(1000, 8), (2000, 8), (1200, 8), (8000, 12), (500, 4)])
def test_host_tables_match_oracle(oracle, nf, nl):
    t = oa.host_tables(nf, 1.2, nl)
    p = oracle.params(nf, 1.2, nl)
    assert list(t["quota"]) == list(p.quota)[:nl]
    assert list(t["umax"]) == list(p.umax)
    for key, ref in (("scale", p.scale), ("inv_scale", p.inv_scale), ("sigma2", p.sigma2), ("inv_sigma2", p.inv_sigma2)):
        assert t[key].tobytes() == np.array(list(ref)[:nl], np.float32).tobytes(), key


@pytest.mark.parametrize("w,h,nf,nl", [(640, 480, 1000, 8), (1241, 376, 2000, 8), (752, 480, 1200, 8),
                                       (3840, 2160, 8000, 12)])
def test_level_geometry(oracle, w, h, nf, nl):
    p = oracle.params(nf, 1.2, nl)
    cells_total = visited = 0
    for l in range(nl):
        g = oa.host_level_geometry(nf, 1.2, nl, w, h, l)
        assert (g.width, g.height) == oracle.level_size(p, w, h, l)
        # reference src/ORBextractor.cc:871-877 and :576 restated in float32
        fw, fh = np.float32(g.width - 32), np.float32(g.height - 32)
        assert g.n_cols == int(fw / np.float32(30)) and g.n_rows == int(fh / np.float32(30))
        assert g.w_cell == int(np.ceil(fw / np.float32(g.n_cols))) and g.h_cell == int(np.ceil(fh / np.float32(g.n_rows)))
        assert g.n_ini == int(np.floor(float(fw / fh) + 0.5))
        assert g.quota == p.quota[l]
        cells_total += g.n_cols * g.n_rows
        visited += g.n_cells
        assert g.n_cells <= g.n_cols * g.n_rows
    expect = {(640, 480): 815, (1241, 376): 1231, (752, 480): 982, (3840, 2160): 27939}[(w, h)]
    assert cells_total == expect
    # KITTI level 0: 40 columns of 31 px overshoot the 1209-px rect, so the last column is never visited (:894-895)
    assert visited == {(640, 480): 815, (1241, 376): 1220, (752, 480): 982}.get((w, h), visited)


def test_geometry_errors():
    lib = oa.load_library()
    g = oa.LevelGeom()
    assert lib.orbx_host_level_geometry(500, 1.2, 8, 100, 100, 0, C.byref(g)) == -5  # levels too small
    assert lib.orbx_host_level_geometry(500, 1.2, 8, 640, 480, 9, C.byref(g)) == -1
    assert lib.orbx_host_level_geometry(500, 1.2, 40, 640, 480, 0, C.byref(g)) == -1
    assert lib.orbx_host_level_geometry(500, 1.2, 2, 100, 400, 0, C.byref(g)) == -5   # tall image: zero quadtree roots
    assert lib.orbx_host_max_keypoints(2000, 1.2, 8, 1241, 376) >= 2000 + 8


@pytest.mark.parametrize("sn,dn", [(1241, 1034), (376, 313), (179, 149), (100, 130), (64, 53)])
def test_resize_taps_reproduce_the_oracle_resize(oracle, sn, dn):
    """A 1-D image resized with the library's taps equals the oracle's resize of the same data."""
    rng = np.random.default_rng(sn + dn)
    row = rng.integers(0, 256, (1, sn), dtype=np.uint8)
    ofs, c0, c1 = oa.host_resize_taps(sn, dn, True)
    assert ofs.min() >= 0 and ofs.max() <= sn - 1
    S = row[0].astype(np.int64)
    T = S[ofs] * c0 + S[np.minimum(ofs + 1, sn - 1)] * c1
    got = ((((2048 * (T >> 4)) >> 16) + ((0 * (T >> 4)) >> 16) + 2) >> 2).astype(np.uint8)
    exp = oracle.resize_linear(np.repeat(row, 4, 0), dn, 4)[0]  # identical rows: vertical taps do nothing
    assert (got == exp).all()
    ofs_v, b0, b1 = oa.host_resize_taps(sn, dn, False)
    assert (b0 == c0).sum() >= dn - 2 and (b0.astype(int) + b1 >= 2047).all()
    assert ofs_v.min() >= -1


def test_create_without_gpu_reports_no_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(oa.OrbxError, match="no usable GPU"):
        oa.ORBextractor(1000, 1.2, 8, 20, 7)


def test_cpp_class_getters(oracle):
    exe = os.path.join(ROOT, "tests", "cpp", "orbextractor_harness")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    out = subprocess.check_output([exe, "tables", "2000", "1.2", "8"]).decode().split("\n")
    assert out[0].startswith("levels 8 scale 1.2")
    p = oracle.params(2000, 1.2, 8)
    for l in range(8):
        vals = [np.float32(v) for v in out[1 + l].split()]
        assert vals == [np.float32(p.scale[l]), np.float32(p.inv_scale[l]), np.float32(p.sigma2[l]),
                        np.float32(p.inv_sigma2[l])]


@pytest.mark.parametrize("w", list(range(62, 140)) + [317, 640, 752, 1241, 1242, 1243, 1244, 3840, 4127])
def test_blur_rim_selectors_reproduce_reflect_101(w):
    """The blur kernel rebuilds the neighbour dwords of a row's first and last two dword columns (and the last dword's
    bytes beyond the width) with byte permutations.  Here the selectors the library computes are applied, in
    numpy, to a row with arbitrary bytes in its padding, and every pixel's 7-tap row sum is compared with the sum
    over cv::GaussianBlur's BORDER_REFLECT_101 columns (reference call site src/ORBextractor.cc:1188)."""
    L = oa.load_library()
    taps = [18, 34, 49, 55, 49, 34, 18]
    rng = np.random.default_rng(w)
    pitch = (w + 63) // 64 * 64
    row = rng.integers(0, 256, pitch).astype(np.int64)       # the padding holds arbitrary bytes
    nd = (w + 3) // 4

    def reflect(p):
        p = -p if p < 0 else p
        return 2 * w - 2 - p if p >= w else p

    def dword(d):                                             # a neighbour outside the row belongs to another tile
        return [-10**6] * 4 if d < 0 or d >= nd else list(row[4 * d:4 * d + 4])

    def perm(hi, lo, sel):
        both = list(lo) + list(hi)
        return [both[(sel >> (8 * k)) & 7] for k in range(4)]

    for dcol in [0, 1, nd - 3, nd - 2, nd - 1]:
        x0 = 4 * dcol
        c, lw, r = dword(dcol), dword(dcol - 1), dword(dcol + 1)
        last = dcol == nd - 1
        sel_l = L.orbx_host_blur_rim_selector(x0 - 4, x0, x0 - 4, w)
        sel_c = L.orbx_host_blur_rim_selector(x0, x0, x0 - 4, w)
        sel_r = L.orbx_host_blur_rim_selector(x0 + 4, x0 - 4 if last else x0 + 4, x0, w)
        z = lw if last else r
        window = perm(c, lw, sel_l) + perm(c, lw, sel_c) + perm(z, c, sel_r)   # columns x0-4 .. x0+7
        for k in range(4):
            x = x0 + k
            if x < w:
                got = sum(t * window[4 + k - 3 + i] for i, t in enumerate(taps))
                assert got == sum(t * row[reflect(x - 3 + i)] for i, t in enumerate(taps)), (w, dcol, k)


@pytest.mark.parametrize("scale", [1.1, 1.2, 1.25, 1.3, 1.5, 2.0])
def test_a_source_row_completes_at_most_two_destination_rows(scale):
    """The streaming resize kernel emits the destination rows whose second tap is the source row just filtered
    with two conditional blocks, not a loop: for every down-scaling level of a pyramid no three destination rows
    may share their second source row (two do where the taps are clamped to the image's last row)."""
    for sn in list(range(62, 400, 7)) + [480, 752, 1241, 2160, 3840]:
        dn = int(np.rint(np.float32(sn) * (np.float32(1.0) / np.float32(scale))))
        if dn < 8:
            continue
        ofs, c0, c1 = oa.host_resize_taps(sn, dn, False)
        ofs1 = np.minimum(ofs + 1, sn - 1)
        assert (ofs1[2:] != ofs1[:-2]).all(), (sn, dn)
