"""ctypes view of oracle/liborb_oracle.so (the CPU restatement; test infrastructure only).

Imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg; never by
the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB_PATH = os.path.join(_ROOT, "oracle", "liborb_oracle.so")

MAX_LEVELS = 16


class OrcKp(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32), ("class_id", C.c_int32)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
CAND_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("response", "<i4")])


class OrcParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int),
                ("scale", C.c_float * MAX_LEVELS), ("inv_scale", C.c_float * MAX_LEVELS),
                ("sigma2", C.c_float * MAX_LEVELS), ("inv_sigma2", C.c_float * MAX_LEVELS),
                ("quota", C.c_int * MAX_LEVELS), ("umax", C.c_int * 16)]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_ROOT, "oracle", "orb_oracle.cpp")):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_libs = {}


def lib(path=None):
    """The oracle library: the committed build, or another build of the same source at `path`."""
    if path not in _libs:
        if path is None and not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(path or _LIB_PATH)
        u8p = C.POINTER(C.c_uint8)
        L.orc_params_init.argtypes = [C.POINTER(OrcParams), C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_level_size.argtypes = [C.POINTER(OrcParams), C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_resize_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_gaussian_blur7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_level_candidates.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_int]
        L.orc_distribute_octree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_int]
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_ic_angle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.orc_ic_angle.restype = C.c_float
        L.orc_sincos_deg.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_descriptor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.orc_pattern.restype = C.POINTER(C.c_int8)
        L.orc_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_create.restype = C.c_void_p
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_get_params.argtypes = [C.c_void_p]
        L.orc_get_params.restype = C.POINTER(OrcParams)
        L.orc_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_num_keypoints.argtypes = [C.c_void_p]
        L.orc_get_keypoints.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_descriptors.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_level_dims.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_get_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_get_blurred.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_get_num_candidates.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_get_num_level_keypoints.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_level_padded.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_undistort_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_assign_grid.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_cvt_gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_remap_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_stereo_match.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_search_window.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                        C.c_int, C.c_int, C.c_float, C.c_void_p, C.POINTER(C.c_int)]
        L.orc_search_by_projection.argtypes = [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_voc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_voc_create.restype = C.c_void_p
        L.orc_voc_destroy.argtypes = [C.c_void_p]
        L.orc_voc_destroy.restype = None
        L.orc_voc_num_words.argtypes = [C.c_void_p]
        L.orc_voc_transform_feature.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_voc_transform_feature.restype = None
        L.orc_voc_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_voc_transform.restype = None
        _ = u8p
        _libs[path] = L
    return _libs[path]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def params(nfeatures, scale_factor, nlevels, ini_th=20, min_th=7):
    p = OrcParams()
    rc = lib().orc_params_init(C.byref(p), nfeatures, scale_factor, nlevels, ini_th, min_th)
    assert rc == 0
    return p


def level_size(p, w0, h0, level):
    w, h = C.c_int(), C.c_int()
    lib().orc_level_size(C.byref(p), w0, h0, level, C.byref(w), C.byref(h))
    return w.value, h.value


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.empty((dh, dw), dtype=np.uint8)
    lib().orc_resize_linear(_ptr(src), src.shape[1], src.shape[0], src.shape[1], _ptr(dst), dw, dh, dw)
    return dst


def gaussian_blur7(src):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.empty_like(src)
    lib().orc_gaussian_blur7(_ptr(src), src.shape[1], src.shape[0], src.shape[1], _ptr(dst), src.shape[1])
    return dst


def fast(img, threshold):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    cap = img.size
    out = np.zeros(cap, dtype=CAND_DTYPE)
    n = lib().orc_fast(_ptr(img), img.shape[1], img.shape[0], img.shape[1], threshold, _ptr(out), cap)
    return out[:n]


def level_candidates(img, ini_th=20, min_th=7):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    cap = img.size
    out = np.zeros(cap, dtype=CAND_DTYPE)
    n = lib().orc_level_candidates(_ptr(img), img.shape[1], img.shape[0], img.shape[1], ini_th, min_th,
                                   _ptr(out), cap)
    return out[:n]


def distribute_octree(cands, minX, maxX, minY, maxY, N):
    cands = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
    cap = len(cands) + 8
    out = np.zeros(cap, dtype=np.int32)
    m = lib().orc_distribute_octree(_ptr(cands), len(cands), minX, maxX, minY, maxY, N, _ptr(out), cap)
    assert m >= 0, m
    return out[:m]


def fast_atan2(y, x):
    return float(lib().orc_fast_atan2(float(y), float(x)))


def ic_angle(img, x, y, umax):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    um = (C.c_int * 16)(*umax)
    return float(lib().orc_ic_angle(_ptr(img), img.shape[1], x, y, um))


def sincos_deg(angle):
    c, s = C.c_float(), C.c_float()
    lib().orc_sincos_deg(float(angle), C.byref(c), C.byref(s))
    return c.value, s.value


def descriptor(blurred, x, y, angle):
    blurred = np.ascontiguousarray(blurred, dtype=np.uint8)
    d = np.zeros(32, dtype=np.uint8)
    lib().orc_descriptor(_ptr(blurred), blurred.shape[1], x, y, float(angle), _ptr(d))
    return d


def pattern():
    p = lib().orc_pattern()
    return np.ctypeslib.as_array(p, shape=(1024,)).copy()


def cvt_gray(src, blue_idx):
    """src: h x w x (3|4) uint8; blue_idx 0 = BGR(A), 2 = RGB(A)."""
    src = np.ascontiguousarray(src, np.uint8)
    h, w, ch = src.shape
    dst = np.zeros((h, w), np.uint8)
    lib().orc_cvt_gray(_ptr(src), w, h, src.strides[0], ch, blue_idx, _ptr(dst), w)
    return dst


def remap_linear(src, mapx, mapy):
    src = np.ascontiguousarray(src, np.uint8)
    mapx = np.ascontiguousarray(mapx, np.float32)
    mapy = np.ascontiguousarray(mapy, np.float32)
    dh, dw = mapx.shape
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_remap_linear(_ptr(src), src.shape[1], src.shape[0], src.strides[0], _ptr(mapx), _ptr(mapy), dw, _ptr(dst),
                           dw, dh, dw)
    return dst


def undistort_points(xy, K, dist):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float32)
    dist = np.ascontiguousarray(dist, np.float32)
    assert K.size == 4 and dist.size == 5
    out = np.zeros_like(xy)
    lib().orc_undistort_points(_ptr(xy), len(xy), _ptr(K), _ptr(dist), _ptr(out))
    return out


def assign_grid(xy, min_x, min_y, inv_w, inv_h):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    start = np.zeros(64 * 48 + 1, np.int32)
    idx = np.zeros(max(1, len(xy)), np.int32)
    lib().orc_assign_grid(_ptr(xy), len(xy), min_x, min_y, inv_w, inv_h, _ptr(start), _ptr(idx))
    return start, idx[:start[-1]]


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    assert a.size == 32 and b.size == 32
    return int(lib().orc_descriptor_distance(_ptr(a), _ptr(b)))


def stereo_match(left, right, mb, mbf):
    """Frame::ComputeStereoMatches on the last extraction of two Oracle instances.
    Returns (u_right, depth, sad, kept)."""
    n = lib().orc_num_keypoints(left._c)
    u = np.zeros(n, np.float32)
    d = np.zeros(n, np.float32)
    sad = np.zeros(n, np.int32)
    kept = lib().orc_stereo_match(left._c, right._c, mb, mbf, _ptr(u), _ptr(d), _ptr(sad))
    return u, d, sad, kept


class Oracle:
    """Whole-path CPU extractor with access to every intermediate."""

    def __init__(self, nfeatures, scale_factor, nlevels, ini_th=20, min_th=7, lib_path=None):
        self._L = lib(lib_path)
        self._c = self._L.orc_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        assert self._c
        self.nlevels = nlevels

    def __del__(self):
        if getattr(self, "_c", None):
            self._L.orc_destroy(self._c)
            self._c = None

    @property
    def params(self):
        return self._L.orc_get_params(self._c).contents

    def extract(self, img):
        img = np.asarray(img)
        assert img.dtype == np.uint8 and img.ndim == 2 and img.strides[1] == 1
        n = self._L.orc_extract(self._c, _ptr(img), img.shape[1], img.shape[0], img.strides[0])
        assert n >= 0, n
        kps = np.zeros(n, dtype=KP_DTYPE)
        desc = np.zeros((n, 32), dtype=np.uint8)
        self._L.orc_get_keypoints(self._c, _ptr(kps))
        self._L.orc_get_descriptors(self._c, _ptr(desc))
        return kps, desc

    def level_dims(self, l):
        w, h = C.c_int(), C.c_int()
        self._L.orc_get_level_dims(self._c, l, C.byref(w), C.byref(h))
        return w.value, h.value

    def level(self, l):
        w, h = self.level_dims(l)
        a = np.empty((h, w), dtype=np.uint8)
        self._L.orc_get_level(self._c, l, _ptr(a))
        return a

    def blurred(self, l):
        w, h = self.level_dims(l)
        a = np.empty((h, w), dtype=np.uint8)
        self._L.orc_get_blurred(self._c, l, _ptr(a))
        return a

    def padded_level(self, l):
        w, h = self.level_dims(l)
        a = np.empty((h + 38, w + 38), dtype=np.uint8)
        self._L.orc_get_level_padded(self._c, l, _ptr(a))
        return a

    def candidates(self, l):
        n = self._L.orc_get_num_candidates(self._c, l)
        a = np.zeros(n, dtype=CAND_DTYPE)
        self._L.orc_get_candidates(self._c, l, _ptr(a))
        return a

    def level_keypoint_count(self, l):
        return self._L.orc_get_num_level_keypoints(self._c, l)


def search_window(xy, octave, desc, u_right, taken, cell_start, cell_idx, min_x, min_y, inv_w, inv_h, u, v, radius,
                  min_level, max_level, ur, q_desc):
    """Frame::GetFeaturesInArea + the best-distance loop of SearchByProjection; returns (index, distance)."""
    xy = np.ascontiguousarray(xy, np.float32)
    octave = np.ascontiguousarray(octave, np.int32)
    desc = np.ascontiguousarray(desc, np.uint8)
    cell_start = np.ascontiguousarray(cell_start, np.int32)
    cell_idx = np.ascontiguousarray(cell_idx, np.int32)
    q_desc = np.ascontiguousarray(q_desc, np.uint8)
    ur_a = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    tk_a = None if taken is None else np.ascontiguousarray(taken, np.uint8)
    d = C.c_int(0)
    i = lib().orc_search_window(_ptr(xy), _ptr(octave), _ptr(desc), None if ur_a is None else _ptr(ur_a),
                                None if tk_a is None else _ptr(tk_a), _ptr(cell_start), _ptr(cell_idx), min_x, min_y,
                                inv_w, inv_h, u, v, radius, min_level, max_level, ur, _ptr(q_desc), C.byref(d))
    return i, d.value


MAP_QUERY_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("octave", "<i4"), ("angle", "<f4"), ("flags", "<i4"),
                            ("desc", "u1", (32,))])
PROJ_PARAMS_DTYPE = np.dtype([(k, "<f4") for k in ("fx", "fy", "cx", "cy", "mbf", "mb", "min_x", "min_y", "max_x", "max_y",
                                                   "inv_w", "inv_h", "th")] + [("mono", "<i4"), ("check_orientation", "<i4")])


def search_by_projection(xy, octave, angle, desc, u_right, taken0, cell_start, cell_idx, scale_factors, params, Tcw, Tlw,
                         queries):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) as the reference runs it, one map
    point after the other; returns (nmatches, match per feature, feature per query)."""
    xy = np.ascontiguousarray(xy, np.float32)
    octave = np.ascontiguousarray(octave, np.int32)
    angle = np.ascontiguousarray(angle, np.float32)
    desc = np.ascontiguousarray(desc, np.uint8)
    cell_start = np.ascontiguousarray(cell_start, np.int32)
    cell_idx = np.ascontiguousarray(cell_idx, np.int32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    P = np.ascontiguousarray(params, PROJ_PARAMS_DTYPE)
    Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(12)
    Tlw = np.ascontiguousarray(Tlw, np.float32).reshape(12)
    q = np.ascontiguousarray(queries, MAP_QUERY_DTYPE)
    ur_a = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    tk_a = None if taken0 is None else np.ascontiguousarray(taken0, np.uint8)
    n_kp = len(octave)
    match = np.full(max(n_kp, 1), -1, np.int32)
    qmatch = np.full(max(len(q), 1), -1, np.int32)
    n = lib().orc_search_by_projection(_ptr(xy), _ptr(octave), _ptr(angle), _ptr(desc), None if ur_a is None else _ptr(ur_a),
                                       None if tk_a is None else _ptr(tk_a), n_kp, _ptr(cell_start), _ptr(cell_idx), _ptr(sf),
                                       _ptr(P), _ptr(Tcw), _ptr(Tlw), _ptr(q), len(q), _ptr(match), _ptr(qmatch))
    return n, match[:n_kp], qmatch[:len(q)]


LOCAL_QUERY_DTYPE = np.dtype([("u", np.float32), ("v", np.float32), ("ur", np.float32), ("level", np.int32),
                              ("view_cos", np.float32), ("flags", np.int32), ("desc", np.uint8, (32,))])
assert LOCAL_QUERY_DTYPE.itemsize == 56


def search_local_map(xy, octave, desc, u_right, taken0, cell_start, cell_idx, scale_factors, min_x, min_y, inv_w, inv_h, th,
                     nn_ratio, queries):
    """ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) as the reference runs it, one map
    point after the other; returns (nmatches, map point per feature, feature per query)."""
    L = lib()
    L.orc_search_local_map.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    xy = np.ascontiguousarray(xy, np.float32)
    octave = np.ascontiguousarray(octave, np.int32)
    desc = np.ascontiguousarray(desc, np.uint8)
    cell_start = np.ascontiguousarray(cell_start, np.int32)
    cell_idx = np.ascontiguousarray(cell_idx, np.int32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    q = np.ascontiguousarray(queries, LOCAL_QUERY_DTYPE)
    ur_a = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    tk_a = None if taken0 is None else np.ascontiguousarray(taken0, np.uint8)
    n_kp = len(octave)
    match = np.full(max(n_kp, 1), -1, np.int32)
    qmatch = np.full(max(len(q), 1), -1, np.int32)
    n = L.orc_search_local_map(_ptr(xy), _ptr(octave), _ptr(desc), None if ur_a is None else _ptr(ur_a),
                               None if tk_a is None else _ptr(tk_a), n_kp, _ptr(cell_start), _ptr(cell_idx), _ptr(sf),
                               min_x, min_y, inv_w, inv_h, th, nn_ratio, _ptr(q), len(q), _ptr(match), _ptr(qmatch))
    return n, match[:n_kp], qmatch[:len(q)]


GUIDED_QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("angle", "<f4"), ("level", "<i4"), ("radius", "<f4"), ("flags", "<i4"),
                               ("desc", "u1", (32,))])
assert GUIDED_QUERY_DTYPE.itemsize == 56


def search_guided(xy, octave, angle, desc, taken0, cell_start, cell_idx, min_x, min_y, inv_w, inv_h, queries, level_up, max_dist,
                  check_orientation):
    """The matching loops of SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (level_up 1) and
    SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (level_up 0) on projected map points (GUIDED_QUERY_DTYPE), one
    after the other; returns (nmatches, map point per feature, feature per query)."""
    L = lib()
    vp = C.c_void_p
    L.orc_search_guided.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, vp, C.c_int,
                                    C.c_int, C.c_int, C.c_int, vp, vp]
    xy = np.ascontiguousarray(xy, np.float32)
    octave = np.ascontiguousarray(octave, np.int32)
    angle = np.ascontiguousarray(angle, np.float32)
    desc = np.ascontiguousarray(desc, np.uint8)
    cell_start = np.ascontiguousarray(cell_start, np.int32)
    cell_idx = np.ascontiguousarray(cell_idx, np.int32)
    q = np.ascontiguousarray(queries, GUIDED_QUERY_DTYPE)
    tk_a = None if taken0 is None else np.ascontiguousarray(taken0, np.uint8)
    n_kp = len(octave)
    match = np.full(max(n_kp, 1), -1, np.int32)
    qmatch = np.full(max(len(q), 1), -1, np.int32)
    n = L.orc_search_guided(_ptr(xy), _ptr(octave), _ptr(angle), _ptr(desc), n_kp, None if tk_a is None else _ptr(tk_a),
                            _ptr(cell_start), _ptr(cell_idx), min_x, min_y, inv_w, inv_h, _ptr(q), len(q), int(level_up),
                            int(max_dist), int(check_orientation), _ptr(match), _ptr(qmatch))
    return n, match[:n_kp], qmatch[:len(q)]


def stereo_from_rgbd(depth, factor, kps_xy, un_x, mbf):
    """Frame::ComputeStereoFromRGBD on a 2-D depth map (uint16 or float32) that GrabImageRGBD would scale by `factor`;
    returns (mvuRight, mvDepth)."""
    L = lib()
    L.orc_stereo_from_rgbd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    L.orc_stereo_from_rgbd.restype = None
    depth = np.ascontiguousarray(depth)
    assert depth.dtype in (np.uint16, np.float32) and depth.ndim == 2
    xy = np.ascontiguousarray(kps_xy, np.float32)
    ux = np.ascontiguousarray(un_x, np.float32)
    n = len(ux)
    ur, z = np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.float32)
    L.orc_stereo_from_rgbd(_ptr(depth), int(depth.dtype == np.float32), depth.shape[1], depth.shape[0], depth.strides[0],
                           factor, _ptr(xy), _ptr(ux), n, mbf, _ptr(ur), _ptr(z))
    return ur[:n], z[:n]


def search_by_bow(fv_node_kf, fv_idx_kf, desc_kf, angle_kf, valid_kf, fv_node_f, fv_idx_f, desc_f, angle_f, nn_ratio,
                  check_orientation):
    """ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) (reference src/ORBmatcher.cc:159-290) on FeatureVectors
    given as (node, feature index) pairs; returns (nmatches, keyframe feature per frame feature)."""
    L = lib()
    L.orc_search_by_bow.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    a = [np.ascontiguousarray(x, t) for x, t in ((fv_node_kf, np.uint32), (fv_idx_kf, np.uint32), (desc_kf, np.uint8),
                                                  (angle_kf, np.float32), (valid_kf, np.uint8), (fv_node_f, np.uint32),
                                                  (fv_idx_f, np.uint32), (desc_f, np.uint8), (angle_f, np.float32))]
    n_f = len(a[8])
    match = np.full(max(n_f, 1), -1, np.int32)
    n = L.orc_search_by_bow(_ptr(a[0]), _ptr(a[1]), len(a[0]), _ptr(a[2]), _ptr(a[3]), _ptr(a[4]), _ptr(a[5]), _ptr(a[6]),
                            len(a[5]), _ptr(a[7]), _ptr(a[8]), n_f, nn_ratio, int(check_orientation), _ptr(match))
    return n, match[:n_f]


def search_by_bow_keyframes(fv_node_1, fv_idx_1, desc_1, angle_1, valid_1, fv_node_2, fv_idx_2, desc_2, angle_2, valid_2,
                            nn_ratio, check_orientation):
    """ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (reference src/ORBmatcher.cc:551-684); returns (nmatches,
    feature of the second keyframe per feature of the first)."""
    L = lib()
    L.orc_search_by_bow_keyframes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_float, C.c_int, C.c_void_p]
    a = [np.ascontiguousarray(x, t) for x, t in ((fv_node_1, np.uint32), (fv_idx_1, np.uint32), (desc_1, np.uint8),
                                                  (angle_1, np.float32), (valid_1, np.uint8), (fv_node_2, np.uint32),
                                                  (fv_idx_2, np.uint32), (desc_2, np.uint8), (angle_2, np.float32),
                                                  (valid_2, np.uint8))]
    n_1, n_2 = len(a[3]), len(a[8])
    match = np.full(max(n_1, 1), -1, np.int32)
    n = L.orc_search_by_bow_keyframes(_ptr(a[0]), _ptr(a[1]), len(a[0]), _ptr(a[2]), _ptr(a[3]), _ptr(a[4]), n_1, _ptr(a[5]),
                                      _ptr(a[6]), len(a[5]), _ptr(a[7]), _ptr(a[8]), _ptr(a[9]), n_2, nn_ratio,
                                      int(check_orientation), _ptr(match))
    return n, match[:n_1]


def bow_score_l1(id1, v1, id2, v2):
    """DBoW2 L1Scoring::score (reference Thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-68) of two BowVectors given as ascending
    word ids with their values."""
    L = lib()
    L.orc_bow_score_l1.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_bow_score_l1.restype = C.c_double
    a, b = np.ascontiguousarray(id1, np.uint32), np.ascontiguousarray(id2, np.uint32)
    x, y = np.ascontiguousarray(v1, np.float64), np.ascontiguousarray(v2, np.float64)
    assert len(a) == len(x) and len(b) == len(y)
    return L.orc_bow_score_l1(_ptr(a), _ptr(x), len(a), _ptr(b), _ptr(y), len(b))


def fuse_search(xy, octave, desc, u_right, cell_start, cell_idx, min_x, min_y, inv_w, inv_h, u, v, ur, radius, level,
                inv_level_sigma2, q_desc):
    """The search of ORBmatcher::Fuse(pKF, vpMapPoints, th) for one projected map point (reference src/ORBmatcher.cc:929-986);
    returns (bestIdx, bestDist)."""
    L = lib()
    vp = C.c_void_p
    L.orc_fuse_search.argtypes = [vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_float, C.c_float, C.c_int, vp, vp, C.POINTER(C.c_int)]
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    octave = np.ascontiguousarray(octave, np.int32)
    desc = np.ascontiguousarray(desc, np.uint8)
    ur_arr = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    cs = np.ascontiguousarray(cell_start, np.int32)
    ci = np.ascontiguousarray(cell_idx, np.int32)
    s2 = np.ascontiguousarray(inv_level_sigma2, np.float32)
    qd = np.ascontiguousarray(q_desc, np.uint8)
    best = C.c_int(0)
    idx = L.orc_fuse_search(_ptr(xy), _ptr(octave), _ptr(desc), None if ur_arr is None else _ptr(ur_arr), _ptr(cs), _ptr(ci),
                            min_x, min_y, inv_w, inv_h, u, v, ur, radius, int(level), _ptr(s2), _ptr(qd), C.byref(best))
    return idx, best.value


def search_for_triangulation(side_1, side_2, f12, epipole, scale_factors, level_sigma2, only_stereo, check_orientation):
    """ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:686-852).  side_k = (fv_node, fv_idx, desc, xy, angle,
    octave, has_mp, u_right) of keyframe k (u_right None: all -1); returns (nmatches, vMatches12)."""
    L = lib()
    vp = C.c_void_p
    L.orc_search_for_triangulation.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp,
                                               vp, C.c_int, vp, C.c_float, C.c_float, vp, vp, C.c_int, C.c_int, vp]

    def prep(side):
        node, idx, desc, xy, angle, octave, has_mp, u_right = side
        n = len(angle)
        ur = np.full(n, -1.0, np.float32) if u_right is None else np.ascontiguousarray(u_right, np.float32)
        return (np.ascontiguousarray(node, np.uint32), np.ascontiguousarray(idx, np.uint32), np.ascontiguousarray(desc, np.uint8),
                np.ascontiguousarray(xy, np.float32).reshape(-1, 2), np.ascontiguousarray(angle, np.float32),
                np.ascontiguousarray(octave, np.int32), np.ascontiguousarray(has_mp, np.uint8), ur, n)
    a, b = prep(side_1), prep(side_2)
    F = np.ascontiguousarray(f12, np.float32).reshape(9)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    s2 = np.ascontiguousarray(level_sigma2, np.float32)
    match = np.full(max(a[8], 1), -1, np.int32)
    n = L.orc_search_for_triangulation(_ptr(a[0]), _ptr(a[1]), len(a[0]), _ptr(a[2]), _ptr(a[3]), _ptr(a[4]), _ptr(a[6]), _ptr(a[7]),
                                       a[8], _ptr(b[0]), _ptr(b[1]), len(b[0]), _ptr(b[2]), _ptr(b[3]), _ptr(b[4]), _ptr(b[5]),
                                       _ptr(b[6]), _ptr(b[7]), b[8], _ptr(F), float(epipole[0]), float(epipole[1]), _ptr(sf),
                                       _ptr(s2), int(only_stereo), int(check_orientation), _ptr(match))
    return n, match[:a[8]]


def search_for_initialization(octave_1, angle_1, desc_1, prev_matched, xy_2, octave_2, angle_2, desc_2, cell_start_2,
                              cell_idx_2, min_x, min_y, inv_w, inv_h, window_size, nn_ratio, check_orientation):
    """ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:418-549); returns (nmatches, vnMatches12,
    the updated vbPrevMatched as an [n1][2] array)."""
    L = lib()
    L.orc_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                                C.c_float, C.c_float, C.c_int, C.c_float, C.c_int, C.c_void_p]
    o1 = np.ascontiguousarray(octave_1, np.int32)
    a1 = np.ascontiguousarray(angle_1, np.float32)
    d1 = np.ascontiguousarray(desc_1, np.uint8)
    pm = np.array(prev_matched, np.float32).reshape(-1, 2).copy()
    xy2 = np.ascontiguousarray(xy_2, np.float32).reshape(-1, 2)
    o2 = np.ascontiguousarray(octave_2, np.int32)
    a2 = np.ascontiguousarray(angle_2, np.float32)
    d2 = np.ascontiguousarray(desc_2, np.uint8)
    cs = np.ascontiguousarray(cell_start_2, np.int32)
    ci = np.ascontiguousarray(cell_idx_2, np.int32)
    n1, n2 = len(o1), len(o2)
    assert len(pm) == n1 and len(cs) == 64 * 48 + 1
    match = np.full(max(n1, 1), -1, np.int32)
    n = L.orc_search_for_initialization(_ptr(o1), _ptr(a1), _ptr(d1), n1, _ptr(pm), _ptr(xy2), _ptr(o2), _ptr(a2), _ptr(d2), n2,
                                        _ptr(cs), _ptr(ci), min_x, min_y, inv_w, inv_h, int(window_size), nn_ratio,
                                        int(check_orientation), _ptr(match))
    return n, match[:n1], pm


class Vocabulary:
    """A DBoW2 vocabulary tree in the oracle (TemplatedVocabulary, reference
    Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h).  Arrays are per node, entry 0 = root."""

    def __init__(self, k, L, scoring, weighting, parent, is_leaf, desc, weight):
        self.parent = np.ascontiguousarray(parent, np.int32)
        self.is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        self.desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        self.weight = np.ascontiguousarray(weight, np.float64)
        n = len(self.parent)
        assert len(self.is_leaf) == n and len(self.desc) == n and len(self.weight) == n
        self.k, self.L, self.scoring, self.weighting = k, L, scoring, weighting
        self._v = lib().orc_voc_create(k, L, scoring, weighting, n, _ptr(self.parent), _ptr(self.is_leaf),
                                       _ptr(self.desc), _ptr(self.weight))
        if not self._v:
            raise ValueError("invalid vocabulary")

    def __del__(self):
        if getattr(self, "_v", None):
            lib().orc_voc_destroy(self._v)
            self._v = None

    @property
    def n_words(self):
        return lib().orc_voc_num_words(self._v)

    def transform_feature(self, desc, levelsup):
        d = np.ascontiguousarray(desc, np.uint8)
        w, nd, wt = C.c_uint32(0), C.c_uint32(0), C.c_double(0)
        lib().orc_voc_transform_feature(self._v, _ptr(d), levelsup, C.byref(w), C.byref(nd), C.byref(wt))
        return w.value, nd.value, wt.value

    def transform(self, descs, levelsup):
        """(bow_id, bow_val, fv_node, fv_idx) of a frame's descriptors (n x 32)."""
        d = np.ascontiguousarray(descs, np.uint8).reshape(-1, 32)
        n = len(d)
        bid, bval = np.zeros(max(1, n), np.uint32), np.zeros(max(1, n), np.float64)
        fnode, fidx = np.zeros(max(1, n), np.uint32), np.zeros(max(1, n), np.uint32)
        nb, nf = C.c_int(0), C.c_int(0)
        lib().orc_voc_transform(self._v, _ptr(d), n, levelsup, _ptr(bid), _ptr(bval), C.byref(nb), _ptr(fnode),
                                _ptr(fidx), C.byref(nf))
        return bid[:nb.value], bval[:nb.value], fnode[:nf.value], fidx[:nf.value]

    def save_text(self, path):
        """The text format of saveToTextFile / loadFromTextFile (:1338-1460)."""
        with open(path, "w") as f:
            f.write("%d %d  %d %d\n" % (self.k, self.L, self.scoring, self.weighting))
            for i in range(1, len(self.parent)):
                f.write("%d %d %s %r\n" % (self.parent[i], int(self.is_leaf[i]), " ".join(str(int(b)) for b in self.desc[i]),
                                            float(self.weight[i])))


def random_vocabulary(k, L, seed, scoring=0, weighting=0, prune=0.0, stop=0.0):
    """A seeded random tree: every node above level L gets k children (a fraction `prune` of the
    inner nodes below the root gets none and so ends early), centres are random bytes, word weights
    uniform in (0.5, 8) with a fraction `stop` of them 0 (stopped words).  Nodes are numbered level
    by level as DBoW2's k-means construction does."""
    rng = np.random.default_rng(seed)
    parent, is_leaf, level = [0], [0], [0]
    frontier = [0]
    for lev in range(1, L + 1):
        nxt = []
        for p in frontier:
            if p != 0 and lev <= L and rng.random() < prune:
                continue  # an inner node without children: a leaf by isLeaf() that the file does not flag
            for _ in range(k):
                parent.append(p)
                is_leaf.append(1 if lev == L else 0)
                level.append(lev)
                nxt.append(len(parent) - 1)
        frontier = nxt
    n = len(parent)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    weight = rng.uniform(0.5, 8.0, n)
    weight[rng.random(n) < stop] = 0.0
    weight[0] = 0.0
    return Vocabulary(k, L, scoring, weighting, parent, is_leaf, desc, weight)


def log_pinned(x):
    """(float32) of the oracle's IEEE-only double logarithm of float32 values."""
    L = lib()
    L.orc_log_pinned.restype = C.c_double
    L.orc_log_pinned.argtypes = [C.c_float]
    return np.array([L.orc_log_pinned(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float64)


def _proj_args(Tcw, Ow, cam, bounds):
    f4 = lambda a, k: (C.c_float * k)(*[float(v) for v in np.asarray(a, np.float32).ravel()])
    return f4(Tcw, 12), f4(Ow, 3), f4(cam, 5), f4(bounds, 4)


def is_in_frustum(Tcw, Ow, cam, bounds, view_cos_limit, log_scale_factor, n_levels, points):
    """Frame::isInFrustum (reference src/Frame.cc:271-327) for the FRUSTUM_POINT_DTYPE-like records `points` (fields x, y, z,
    nx, ny, nz, min_dist, max_dist, flags, desc) of one frame; points with flags bit 1 are skipped as
    Tracking::SearchLocalPoints does (src/Tracking.cc:1343-1346).  Returns LOCAL_QUERY_DTYPE records and nToMatch."""
    L = lib()
    L.orc_is_in_frustum.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                                      C.c_float] + [C.c_void_p] * 5
    T, O, cm, bd = _proj_args(Tcw, Ow, cam, bounds)
    out = np.zeros(len(points), LOCAL_QUERY_DTYPE)
    n_view = 0
    for i, p in enumerate(points):
        out[i]["desc"] = p["desc"]
        out[i]["flags"] = (int(p["flags"]) & 1) | 2
        if int(p["flags"]) & 2:
            continue
        P = (C.c_float * 3)(float(p["x"]), float(p["y"]), float(p["z"]))
        Pn = (C.c_float * 3)(float(p["nx"]), float(p["ny"]), float(p["nz"]))
        u, v, ur, vc = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        lv = C.c_int()
        ok = L.orc_is_in_frustum(T, O, cm, bd, float(view_cos_limit), float(log_scale_factor), int(n_levels), P, Pn,
                                 float(p["min_dist"]), float(p["max_dist"]), C.byref(u), C.byref(v), C.byref(ur),
                                 C.byref(lv), C.byref(vc))
        if ok:
            n_view += 1
            out[i]["u"], out[i]["v"], out[i]["ur"] = u.value, v.value, ur.value
            out[i]["level"], out[i]["view_cos"] = lv.value, vc.value
            out[i]["flags"] = int(p["flags"]) & 1
    return out, n_view


def fuse_project(Tcw, Ow, cam, bounds, log_scale_factor, n_levels, scale_factors, th, points):
    """The tests of ORBmatcher::Fuse in front of its search (reference src/ORBmatcher.cc:881-927) for the records `points`
    of one keyframe; returns (u, v, ur, radius, level, ok) arrays, radius -1 where the point is dropped or skipped."""
    L = lib()
    L.orc_fuse_project.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                                     C.c_float, C.c_float] + [C.c_void_p] * 5
    T, O, cm, bd = _proj_args(Tcw, Ow, cam, bounds)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    n = len(points)
    uo, vo, uro = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    rad, lvl, okv = np.full(n, -1.0, np.float32), np.zeros(n, np.int32), np.zeros(n, bool)
    for i, p in enumerate(points):
        if int(p["flags"]) & 2:
            continue
        P = (C.c_float * 3)(float(p["x"]), float(p["y"]), float(p["z"]))
        Pn = (C.c_float * 3)(float(p["nx"]), float(p["ny"]), float(p["nz"]))
        u, v, ur, r = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        lv = C.c_int()
        if L.orc_fuse_project(T, O, cm, bd, float(log_scale_factor), int(n_levels), sf.ctypes.data, float(th), P, Pn,
                              float(p["min_dist"]), float(p["max_dist"]), C.byref(u), C.byref(v), C.byref(ur), C.byref(r),
                              C.byref(lv)):
            uo[i], vo[i], uro[i], rad[i], lvl[i], okv[i] = u.value, v.value, ur.value, r.value, lv.value, True
    return uo, vo, uro, rad, lvl, okv


def project_guided(Tcw, Ow, cam, bounds, log_scale_factor, n_levels, scale_factors, th, points, angles, relocalisation):
    """The projections in front of the guided searches for the records `points` of one frame: relocalisation False =
    ORBmatcher::SearchByProjection(pKF, Scw, ...) (reference src/ORBmatcher.cc:321-360, the tests of Fuse), True =
    SearchByProjection(CurrentFrame, pKF, ...) (:1534-1563).  Returns GUIDED_QUERY_DTYPE records and the number projected."""
    L = lib()
    L.orc_reloc_project.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_float,
                                                      C.c_float] + [C.c_void_p] * 4
    n = len(points)
    out = np.zeros(n, GUIDED_QUERY_DTYPE)
    out["desc"] = points["desc"]
    out["angle"] = 0.0 if angles is None else np.asarray(angles, np.float32)
    out["flags"], out["radius"] = 3, -1.0
    if not relocalisation:
        u, v, ur, rad, lvl, ok = fuse_project(Tcw, Ow, cam, bounds, log_scale_factor, n_levels, scale_factors, th, points)
        out["u"], out["v"], out["radius"], out["level"] = u, v, rad, lvl
        out["flags"] = np.where(ok, 1, 3)
        return out, int(ok.sum())
    T, O, cm, bd = _proj_args(Tcw, Ow, cam, bounds)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    count = 0
    for i, p in enumerate(points):
        if int(p["flags"]) & 2:
            continue
        P = (C.c_float * 3)(float(p["x"]), float(p["y"]), float(p["z"]))
        u, v, r = C.c_float(), C.c_float(), C.c_float()
        lv = C.c_int()
        if L.orc_reloc_project(T, O, cm, bd, float(log_scale_factor), int(n_levels), sf.ctypes.data, float(th), P,
                               float(p["min_dist"]), float(p["max_dist"]), C.byref(u), C.byref(v), C.byref(r), C.byref(lv)):
            out[i]["u"], out[i]["v"], out[i]["radius"], out[i]["level"], out[i]["flags"] = u.value, v.value, r.value, lv.value, 1
            count += 1
    return out, count


def sim3_project(T1w, S21, cam, bounds, log_scale_factor, n_levels, scale_factors, th, points):
    """The projection of one direction of ORBmatcher::SearchBySim3 (reference src/ORBmatcher.cc:1195-1226) for the records
    `points` of one keyframe pair; returns (u, v, radius, level, ok) arrays, radius -1 where the point is dropped or skipped."""
    L = lib()
    L.orc_sim3_project.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_float,
                                                     C.c_float] + [C.c_void_p] * 4
    f4 = lambda a, k: (C.c_float * k)(*[float(v) for v in np.asarray(a, np.float32).ravel()])
    T, S, cm, bd = f4(T1w, 12), f4(S21, 12), f4(cam, 5), f4(bounds, 4)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    n = len(points)
    uo, vo = np.zeros(n, np.float32), np.zeros(n, np.float32)
    rad, lvl, okv = np.full(n, -1.0, np.float32), np.zeros(n, np.int32), np.zeros(n, bool)
    for i, p in enumerate(points):
        if int(p["flags"]) & 2:
            continue
        P = (C.c_float * 3)(float(p["x"]), float(p["y"]), float(p["z"]))
        u, v, r = C.c_float(), C.c_float(), C.c_float()
        lv = C.c_int()
        if L.orc_sim3_project(T, S, cm, bd, float(log_scale_factor), int(n_levels), sf.ctypes.data, float(th), P,
                              float(p["min_dist"]), float(p["max_dist"]), C.byref(u), C.byref(v), C.byref(r), C.byref(lv)):
            uo[i], vo[i], rad[i], lvl[i], okv[i] = u.value, v.value, r.value, lv.value, True
    return uo, vo, rad, lvl, okv
