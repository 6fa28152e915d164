"""orb_slam2_amd -- MI355X-native ORB feature extraction behind the ORB_SLAM2::ORBextractor API.

The product is `liborbx.so` (hand-written HIP for gfx950 behind the C ABI of include/orbx.h).
This package is the Python host-side mirror of the reference class
(reference include/ORBextractor.h:45-111): same constructor arguments, call operator, getters
and the `mvImagePyramid` member.  There is no CPU fallback: without the library, or without a
GPU, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborbx.so")

MAX_LEVELS = 16
NUM_STAGES = 5
EDGE_THRESHOLD = 19
NOALLOC_SCRATCH_BYTES = 8 << 20

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

DBG_BLURRED, DBG_CANDIDATES, DBG_LEVEL_KPS = 1, 2, 3
FLAG_NO_OVERLAP, FLAG_RESIZE_TILES, FLAG_STEREO_BORDER_PATH, FLAG_TRACE, FLAG_NO_GRAPH, FLAG_PIPE_CHUNK = 1, 2, 3, 4, 5, 6
FLAG_INIT_LIST = 7
FLAG_POISON = 8
FLAG_ORDERED = 9
E_DEVICE_DATA = -7


class OrbxError(RuntimeError):
    pass


class _Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("device_id", C.c_int32),
                ("max_width", C.c_int32), ("max_height", C.c_int32), ("max_batch", C.c_int32),
                ("flags", C.c_uint32)]


class LevelGeom(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("n_cols", C.c_int32), ("n_rows", C.c_int32),
                ("w_cell", C.c_int32), ("h_cell", C.c_int32), ("n_cells", C.c_int32), ("n_ini", C.c_int32),
                ("quota", C.c_int32)]


EXPORTS = [
    "orbx_host_tables", "orbx_host_level_geometry", "orbx_host_resize_taps", "orbx_host_max_keypoints", "orbx_host_blur_rim_selector",
    "orbx_create", "orbx_destroy", "orbx_extract", "orbx_extract_batch", "orbx_extract_batch_device",
    "orbx_synchronize", "orbx_pyramid_level", "orbx_debug_read", "orbx_set_profiling", "orbx_get_stage_ms",
    "orbx_stage_name", "orbx_strerror", "orbx_last_hip_error", "orbx_version",
    "orbx_stereo_match", "orbx_stereo_match_batch_device", "orbx_gray_batch_device", "orbx_remap_batch_device",
    "orbx_undistort_grid_batch_device",
    "orbx_voc_create", "orbx_voc_load_text", "orbx_voc_destroy", "orbx_voc_info", "orbx_voc_node", "orbx_bow_transform_batch_device", "orbx_search_window_batch_device",
    "orbx_search_by_projection_batch_device", "orbx_set_debug_flag",
    "orbx_stereo_match_batch_device_noalloc", "orbx_search_local_map_batch_device", "orbx_search_by_bow_batch_device",
    "orbx_search_by_bow_keyframes_batch_device", "orbx_rgbd_depth_batch_device",
    "orbx_search_for_initialization_batch_device", "orbx_search_for_triangulation_batch_device",
    "orbx_fuse_search_batch_device", "orbx_search_guided_batch_device", "orbx_bow_score_l1_batch_device",
    "orbx_is_in_frustum_batch_device", "orbx_fuse_project_batch_device", "orbx_project_guided_batch_device",
    "orbx_sim3_project_batch_device",
    "orbx_extract_pair", "orbx_set_keep_pyramid", "orbx_padded_level",
    "orbx_host_frame_shard", "orbx_pool_create", "orbx_pool_destroy", "orbx_pool_workers", "orbx_pool_extract_stream",
]

_lib = None


def load_library():
    """Load liborbx.so (raises OrbxError if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("ORBX_LIB", LIB_PATH)  # ORBX_LIB: a diagnostic build of the same library
    if not os.path.exists(path):
        raise OrbxError("liborbx.so not built: run `make -C orb_slam2_amd/csrc` (or __graft_entry__.build())")
    L = C.CDLL(path)
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    L.orbx_host_tables.argtypes = [i32, f32, i32, vp, vp, vp, vp, vp, vp]
    L.orbx_host_level_geometry.argtypes = [i32, f32, i32, i32, i32, i32, C.POINTER(LevelGeom)]
    L.orbx_host_resize_taps.argtypes = [i32, i32, i32, vp, vp, vp]
    L.orbx_host_max_keypoints.argtypes = [i32, f32, i32, i32, i32]
    L.orbx_host_blur_rim_selector.argtypes = [i32, i32, i32, i32]
    L.orbx_host_blur_rim_selector.restype = C.c_uint32
    L.orbx_create.argtypes = [C.POINTER(_Params), C.POINTER(vp)]
    L.orbx_destroy.argtypes = [vp]
    L.orbx_destroy.restype = None
    L.orbx_extract.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, C.POINTER(i32)]
    L.orbx_extract_batch.argtypes = [vp, i32, vp, i32, i32, vp, vp, i32, vp, vp]
    L.orbx_extract_batch_device.argtypes = [vp, i32, vp, i32, i32, i32, C.c_size_t, vp, i32, vp, vp, i32]
    L.orbx_synchronize.argtypes = [vp]
    L.orbx_stereo_match.argtypes = [vp, i32, vp, i32, f32, f32, vp, vp, i32, C.POINTER(i32)]
    L.orbx_stereo_match_batch_device.argtypes = [vp, i32, vp, i32, vp, vp, f32, f32, vp, vp, i32]
    L.orbx_stereo_match_batch_device_noalloc.argtypes = [vp, i32, vp, i32, vp, vp, f32, f32, vp, vp, i32]
    L.orbx_undistort_grid_batch_device.argtypes = [vp, i32, vp, i32, vp, vp, vp, f32, f32, f32, f32, vp, vp, vp, i32]
    L.orbx_voc_create.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, vp, C.POINTER(vp)]
    L.orbx_voc_load_text.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.orbx_voc_destroy.argtypes = [vp]
    L.orbx_voc_destroy.restype = None
    L.orbx_voc_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.orbx_voc_node.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(C.c_uint32), vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.orbx_bow_transform_batch_device.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32]
    L.orbx_search_window_batch_device.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp, f32, f32, f32, f32, vp, i32, vp, vp,
                                                  vp, i32]
    L.orbx_search_by_projection_batch_device.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32]
    L.orbx_search_local_map_batch_device.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, vp, i32, vp,
                                                     vp, vp, vp, i32]
    L.orbx_rgbd_depth_batch_device.argtypes = [vp, i32, vp, i32, i32, i32, i32, C.c_size_t, f32, vp, vp, i32, vp, f32, vp, vp, i32]
    L.orbx_search_by_bow_batch_device.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, f32, i32, vp, vp, i32]
    L.orbx_search_by_bow_keyframes_batch_device.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, f32,
                                                            i32, vp, vp, i32]
    L.orbx_search_for_initialization_batch_device.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, f32, f32, f32,
                                                              f32, i32, f32, i32, vp, vp, i32]
    L.orbx_search_for_triangulation_batch_device.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp,
                                                             i32, vp, vp, i32, i32, vp, vp, i32]
    L.orbx_fuse_search_batch_device.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, f32, f32, f32, f32, vp, i32, vp, vp, vp, i32]
    L.orbx_search_guided_batch_device.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, f32, f32, f32, f32, vp, i32, vp, i32, i32, i32,
                                                  vp, vp, vp, i32]
    L.orbx_bow_score_l1_batch_device.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, i32]
    L.orbx_is_in_frustum_batch_device.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, i32]
    L.orbx_fuse_project_batch_device.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, i32]
    L.orbx_project_guided_batch_device.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, i32]
    L.orbx_sim3_project_batch_device.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, vp, i32]
    L.orbx_gray_batch_device.argtypes = [vp, i32, vp, i32, i32, i32, C.c_size_t, i32, i32, vp, i32, C.c_size_t, i32]
    L.orbx_remap_batch_device.argtypes = [vp, i32, vp, i32, i32, i32, C.c_size_t, vp, vp, i32, vp, i32, i32, i32,
                                          C.c_size_t, i32]
    L.orbx_pyramid_level.argtypes = [vp, i32, i32, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.orbx_extract_pair.argtypes = [vp, vp, i32, vp, i32, i32, i32, vp, i32, vp, vp, f32, f32, vp, vp]
    L.orbx_set_keep_pyramid.argtypes = [vp, i32]
    L.orbx_host_frame_shard.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.orbx_pool_create.argtypes = [C.POINTER(_Params), vp, i32, C.POINTER(vp)]
    L.orbx_pool_destroy.argtypes = [vp]
    L.orbx_pool_destroy.restype = None
    L.orbx_pool_workers.argtypes = [vp]
    L.orbx_pool_extract_stream.argtypes = [vp, i32, vp, i32, i32, vp, vp, i32, vp, vp]
    L.orbx_padded_level.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.orbx_debug_read.argtypes = [vp, i32, i32, i32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.orbx_set_profiling.argtypes = [vp, i32]
    L.orbx_set_debug_flag.argtypes = [vp, i32, i32]
    L.orbx_get_stage_ms.argtypes = [vp, vp]
    L.orbx_stage_name.argtypes = [i32]
    L.orbx_stage_name.restype = C.c_char_p
    L.orbx_strerror.argtypes = [i32]
    L.orbx_strerror.restype = C.c_char_p
    L.orbx_last_hip_error.argtypes = [vp]
    L.orbx_version.restype = C.c_char_p
    _lib = L
    return L


def source_hash():
    """sha256 over the kernel and launcher sources of the library (orb_slam2_amd/csrc, include/orbx.h).  Counter
    summaries under profiles/ carry it, so that numbers collected on other code are recognised as stale."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.h*")) + glob.glob(os.path.join(_HERE, "csrc", "*.cpp")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "orbx.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _check(rc, ctx=None):
    if rc != 0:
        L = load_library()
        msg = L.orbx_strerror(rc).decode()
        if ctx is not None and rc == -2:
            msg += " (hipError %d)" % L.orbx_last_hip_error(ctx)
        raise OrbxError("orbx: %s (%d)" % (msg, rc))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def host_tables(nfeatures, scale_factor, nlevels):
    """Constructor tables of the reference (no GPU needed)."""
    L = load_library()
    f = [np.zeros(nlevels, np.float32) for _ in range(4)]
    quota = np.zeros(nlevels, np.int32)
    umax = np.zeros(16, np.int32)
    _check(L.orbx_host_tables(nfeatures, scale_factor, nlevels, _ptr(f[0]), _ptr(f[1]), _ptr(f[2]), _ptr(f[3]),
                              _ptr(quota), _ptr(umax)))
    return {"scale": f[0], "inv_scale": f[1], "sigma2": f[2], "inv_sigma2": f[3], "quota": quota, "umax": umax}


def host_level_geometry(nfeatures, scale_factor, nlevels, width, height, level):
    g = LevelGeom()
    _check(load_library().orbx_host_level_geometry(nfeatures, scale_factor, nlevels, width, height, level, C.byref(g)))
    return g


def host_resize_taps(sn, dn, clamp_index):
    ofs = np.zeros(dn, np.int32)
    c0 = np.zeros(dn, np.int16)
    c1 = np.zeros(dn, np.int16)
    _check(load_library().orbx_host_resize_taps(sn, dn, int(clamp_index), _ptr(ofs), _ptr(c0), _ptr(c1)))
    return ofs, c0, c1


def host_max_keypoints(nfeatures, scale_factor, nlevels, width, height):
    rc = load_library().orbx_host_max_keypoints(nfeatures, scale_factor, nlevels, width, height)
    if rc < 0:
        _check(rc)
    return rc


def host_frame_shard(n_frames, rank, world):
    """[first, last) of worker `rank` of `world` (orbx_host_frame_shard: contiguous runs of stereo pairs)."""
    a, b = C.c_int(0), C.c_int(0)
    _check(load_library().orbx_host_frame_shard(n_frames, rank, world, C.byref(a), C.byref(b)))
    return a.value, b.value


class ExtractorPool:
    """Several GPUs in one process (orbx_pool_*): one worker thread and one context per listed device; a stream of
    host frames is cut by frame into pair-aligned shards, results come back in frame order."""

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, devices, *, max_width, max_height, max_batch=256):
        self._pool = None
        L = load_library()
        self._L = L
        self.nfeatures, self.nlevels, self.scaleFactor = int(nfeatures), int(nlevels), float(np.float32(scaleFactor))
        p = _Params(self.nfeatures, self.scaleFactor, self.nlevels, int(iniThFAST), int(minThFAST), 0, max_width, max_height,
                    max_batch, 0)
        dev = np.asarray(list(devices), np.int32)
        pool = C.c_void_p()
        _check(L.orbx_pool_create(C.byref(p), _ptr(dev), len(dev), C.byref(pool)))
        self._pool = pool

    def workers(self):
        return self._L.orbx_pool_workers(self._pool)

    def extract_stream(self, images, out=None):
        """images: list of equally sized 2-D uint8 arrays.  Returns (kps [n][cap], desc [n][cap][32], counts [n]);
        `out` = such a triple to reuse."""
        n = len(images)
        h, w = images[0].shape
        cap = host_max_keypoints(self.nfeatures, self.scaleFactor, self.nlevels, w, h)
        kps, desc, cnt = out if out is not None else (np.zeros((n, cap), KP_DTYPE), np.zeros((n, cap, 32), np.uint8),
                                                      np.zeros(n, np.int32))
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in images])
        strides = np.array([im.strides[0] for im in images], np.int32)
        _check(self._L.orbx_pool_extract_stream(self._pool, n, ptrs, w, h, _ptr(strides), _ptr(kps), cap, _ptr(desc), _ptr(cnt)))
        return kps, desc, cnt

    def close(self):
        if self._pool is not None:
            self._L.orbx_pool_destroy(self._pool)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ORBextractor:
    """Mirror of ORB_SLAM2::ORBextractor on the MI355X.

    ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) as in the reference
    (include/ORBextractor.h:51-52); the keyword arguments size the device context.
    """

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, *, device=0,
                 max_width=1280, max_height=1024, max_batch=1, keep_pyramid=True):
        self._ctx = None
        L = load_library()
        self._L = L
        self.nfeatures, self.nlevels = int(nfeatures), int(nlevels)
        self.scaleFactor = float(np.float32(scaleFactor))
        self.iniThFAST, self.minThFAST = int(iniThFAST), int(minThFAST)
        t = host_tables(self.nfeatures, self.scaleFactor, self.nlevels)
        self.mvScaleFactor, self.mvInvScaleFactor = t["scale"], t["inv_scale"]
        self.mvLevelSigma2, self.mvInvLevelSigma2 = t["sigma2"], t["inv_sigma2"]
        self.mnFeaturesPerLevel, self.umax = t["quota"], t["umax"]
        self.mvImagePyramid = [None] * self.nlevels
        self.keep_pyramid = keep_pyramid
        p = _Params(self.nfeatures, self.scaleFactor, self.nlevels, self.iniThFAST, self.minThFAST, device,
                    max_width, max_height, max_batch, 0)
        ctx = C.c_void_p()
        _check(L.orbx_create(C.byref(p), C.byref(ctx)))
        self._ctx = ctx
        self._max = (max_width, max_height, max_batch)
        self._size = (max_width, max_height)
        if keep_pyramid:   # the padded pyramid is built on the device and copied inside the extraction's own submission
            _check(L.orbx_set_keep_pyramid(ctx, EDGE_THRESHOLD), ctx)

    def close(self):
        if self._ctx is not None:
            self._L.orbx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- getters of the reference (include/ORBextractor.h:63-83) ----
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self.scaleFactor

    def GetScaleFactors(self):
        return self.mvScaleFactor.copy()

    def GetInverseScaleFactors(self):
        return self.mvInvScaleFactor.copy()

    def GetScaleSigmaSquares(self):
        return self.mvLevelSigma2.copy()

    def GetInverseScaleSigmaSquares(self):
        return self.mvInvLevelSigma2.copy()

    def max_keypoints(self, width, height):
        return host_max_keypoints(self.nfeatures, self.scaleFactor, self.nlevels, width, height)

    # ---- operator() of the reference (include/ORBextractor.h:59-61) ----
    def __call__(self, image, mask=None):
        """Returns (keypoints[K] structured array, descriptors[K,32] uint8).

        An empty image returns (None, None) -- the reference returns without touching its
        outputs (src/ORBextractor.cc:1140-1141).  `mask` is ignored, as in the reference.
        With no keypoints the descriptor array is empty (reference releases it, :1160-1161).
        """
        if image is None or getattr(image, "size", 0) == 0:
            return None, None
        image = np.asarray(image)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise OrbxError("image must be 8-bit single channel (CV_8UC1)")  # reference asserts, :1144
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        self._size = (w, h)
        cap = self.max_keypoints(w, h)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        _check(self._L.orbx_extract(self._ctx, _ptr(image), w, h, image.strides[0], _ptr(kps), cap, _ptr(desc),
                                    C.byref(n)), self._ctx)
        if self.keep_pyramid:
            self.mvImagePyramid = [self.padded_level(l) for l in range(self.nlevels)]
        return kps[:n.value].copy(), desc[:n.value].copy()

    def padded_level(self, level, copy=True):
        """Level `level` of the last single-image extraction as the reference keeps it in mvImagePyramid: the un-padded
        ROI of a buffer with a 19-px reflect-101 border (src/ORBextractor.cc:1227-1245), built on the device.  copy=False
        returns a view of the context's pinned memory, valid until the next extraction."""
        base, w, h, st = C.c_void_p(), C.c_int(), C.c_int(), C.c_int()
        _check(self._L.orbx_padded_level(self._ctx, level, C.byref(base), C.byref(w), C.byref(h), C.byref(st)), self._ctx)
        E = EDGE_THRESHOLD
        raw = (C.c_uint8 * (st.value * (h.value + 2 * E))).from_address(base.value)
        buf = np.frombuffer(raw, np.uint8).reshape(h.value + 2 * E, st.value)
        if copy:
            buf = buf.copy()
        return buf[E:E + h.value, E:E + w.value]

    def extract_pair(self, left, right, mb=None, mbf=None):
        """A stereo pair through one context as one call (orbx_extract_pair: what the stereo Frame constructor does
        with two extractor instances on two threads and ComputeStereoMatches, src/Frame.cc:78-81, 90).  Returns
        ((kps_l, desc_l), (kps_r, desc_r), u_right, depth); the last two are None when mb is None."""
        left, right = np.ascontiguousarray(left), np.ascontiguousarray(right)
        if left.dtype != np.uint8 or left.ndim != 2 or right.dtype != np.uint8 or right.shape != left.shape:
            raise OrbxError("pair images must be 8-bit single channel and equally sized")
        h, w = left.shape
        self._size = (w, h)
        cap = self.max_keypoints(w, h)
        kps, desc, n = np.zeros((2, cap), KP_DTYPE), np.zeros((2, cap, 32), np.uint8), np.zeros(2, np.int32)
        stereo = mb is not None
        u, d = (np.full(cap, -1, np.float32), np.full(cap, -1, np.float32)) if stereo else (None, None)
        _check(self._L.orbx_extract_pair(self._ctx, _ptr(left), left.strides[0], _ptr(right), right.strides[0], w, h, _ptr(kps),
                                         cap, _ptr(desc), _ptr(n), float(mb or 0), float(mbf or 0),
                                         _ptr(u) if stereo else None, _ptr(d) if stereo else None), self._ctx)
        out = [(kps[i, :n[i]].copy(), desc[i, :n[i]].copy()) for i in range(2)]
        return out[0], out[1], (u[:n[0]] if stereo else None), (d[:n[0]] if stereo else None)

    def extract_batch(self, images):
        """images: (n, h, w) uint8 array or list of equally sized 2-D arrays, host memory."""
        imgs = [np.asarray(im) for im in images]
        n = len(imgs)
        h, w = imgs[0].shape
        for k, im in enumerate(imgs):
            if im.dtype != np.uint8 or im.shape != (h, w):
                raise OrbxError("batch images must be uint8 and equally sized")
            if im.strides[1] != 1:
                imgs[k] = np.ascontiguousarray(im)
        self._size = (w, h)
        cap = self.max_keypoints(w, h)
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        strides = np.array([im.strides[0] for im in imgs], np.int32)
        kps = np.zeros((n, cap), KP_DTYPE)
        desc = np.zeros((n, cap, 32), np.uint8)
        cnt = np.zeros(n, np.int32)
        _check(self._L.orbx_extract_batch(self._ctx, n, ptrs, w, h, _ptr(strides), _ptr(kps), cap, _ptr(desc),
                                          _ptr(cnt)), self._ctx)
        return [(kps[i, :cnt[i]].copy(), desc[i, :cnt[i]].copy()) for i in range(n)]

    def extract_batch_device(self, d_imgs, n, width, height, stride, frame_stride, d_kps, kp_cap, d_desc, d_n,
                             sync=True):
        """All arguments are raw device addresses (ints); see orbx_extract_batch_device."""
        self._size = (width, height)
        _check(self._L.orbx_extract_batch_device(self._ctx, n, C.c_void_p(d_imgs), width, height, stride,
                                                 frame_stride, C.c_void_p(d_kps), kp_cap, C.c_void_p(d_desc),
                                                 C.c_void_p(d_n), int(sync)), self._ctx)

    def synchronize(self):
        _check(self._L.orbx_synchronize(self._ctx), self._ctx)

    # ---- the stage before the extractor: gray conversion and rectification, device memory ----
    def gray_batch_device(self, n, d_src, width, height, src_stride, src_frame_stride, channels, blue_idx, d_dst,
                          dst_stride, dst_frame_stride, sync=True):
        """cv::cvtColor(..., CV_*2GRAY) of Tracking::GrabImage* (src/Tracking.cc:175-199); raw device addresses."""
        _check(self._L.orbx_gray_batch_device(self._ctx, n, C.c_void_p(d_src), width, height, src_stride, src_frame_stride,
                                              channels, blue_idx, C.c_void_p(d_dst), dst_stride, dst_frame_stride,
                                              int(sync)), self._ctx)

    def remap_batch_device(self, n, d_src, src_width, src_height, src_stride, src_frame_stride, d_mapx, d_mapy, map_stride,
                           d_dst, dst_width, dst_height, dst_stride, dst_frame_stride, sync=True):
        """cv::remap(..., INTER_LINEAR) of the stereo examples (Examples/Stereo/stereo_euroc.cc:137); raw device
        addresses."""
        _check(self._L.orbx_remap_batch_device(self._ctx, n, C.c_void_p(d_src), src_width, src_height, src_stride,
                                               src_frame_stride, C.c_void_p(d_mapx), C.c_void_p(d_mapy), map_stride,
                                               C.c_void_p(d_dst), dst_width, dst_height, dst_stride, dst_frame_stride,
                                               int(sync)), self._ctx)

    # ---- right after the extractor: Frame::UndistortKeyPoints + AssignFeaturesToGrid ----
    def undistort_grid_batch_device(self, n, d_kps, kp_cap, d_n, K, dist, min_x, min_y, inv_w, inv_h, d_kps_un,
                                    d_cell_start, d_cell_idx, sync=True):
        """mvKeysUn and mGrid (src/Frame.cc:421-451, 232-247) for every frame of a batch; K = (fx, fy, cx, cy),
        dist = (k1, k2, p1, p2, k3); raw device addresses; see orbx_undistort_grid_batch_device."""
        K = np.ascontiguousarray(K, np.float32)
        dist = np.ascontiguousarray(dist, np.float32)
        assert K.size == 4 and dist.size == 5
        _check(self._L.orbx_undistort_grid_batch_device(self._ctx, n, C.c_void_p(d_kps), kp_cap, C.c_void_p(d_n), _ptr(K),
                                                        _ptr(dist), min_x, min_y, inv_w, inv_h, C.c_void_p(d_kps_un),
                                                        C.c_void_p(d_cell_start), C.c_void_p(d_cell_idx), int(sync)),
               self._ctx)

    # ---- descriptor consumer: Frame::ComputeBoW (src/Frame.cc:411-418) ----
    def bow_transform_batch_device(self, voc, n, d_desc, kp_cap, d_n, levelsup, d_word, d_node, d_weight, d_bow_id,
                                   d_bow_val, d_bow_n, d_fv_node, d_fv_idx, d_fv_n, sync=True):
        """mBowVec and mFeatVec of every frame of a batch (DBoW2's transform with `levelsup`); `voc` is an
        ORBVocabulary; raw device addresses; see orbx_bow_transform_batch_device."""
        _check(self._L.orbx_bow_transform_batch_device(
            self._ctx, voc._v, n, C.c_void_p(d_desc), kp_cap, C.c_void_p(d_n), levelsup, C.c_void_p(d_word),
            C.c_void_p(d_node), C.c_void_p(d_weight), C.c_void_p(d_bow_id), C.c_void_p(d_bow_val), C.c_void_p(d_bow_n),
            C.c_void_p(d_fv_node), C.c_void_p(d_fv_idx), C.c_void_p(d_fv_n), int(sync)), self._ctx)

    def bow_score_l1_batch_device(self, n_pairs, d_bow_id_1, d_bow_val_1, d_bow_n_1, cap_1, d_bow_id_2, d_bow_val_2, d_bow_n_2,
                                  cap_2, d_pairs, d_score, sync=True):
        """ORBVocabulary::score (L1 norm, DBoW2 ScoringObject.cpp:23-68) for a batch of pairs of BowVectors; see
        orbx_bow_score_l1_batch_device."""
        _check(self._L.orbx_bow_score_l1_batch_device(
            self._ctx, n_pairs, C.c_void_p(d_bow_id_1), C.c_void_p(d_bow_val_1), C.c_void_p(d_bow_n_1), cap_1,
            C.c_void_p(d_bow_id_2), C.c_void_p(d_bow_val_2), C.c_void_p(d_bow_n_2), cap_2, C.c_void_p(d_pairs),
            C.c_void_p(d_score), int(sync)), self._ctx)

    def search_window_batch_device(self, n, d_kps_un, kp_cap, d_cell_start, d_cell_idx, d_desc, d_u_right, d_taken, min_x,
                                   min_y, inv_w, inv_h, d_queries, q_cap, d_nq, d_best_idx, d_best_dist, sync=True):
        """Frame::GetFeaturesInArea + the best-distance loop of ORBmatcher::SearchByProjection for q_cap queries
        (WINDOW_QUERY_DTYPE records) per frame; d_u_right / d_taken may be 0; see orbx_search_window_batch_device."""
        _check(self._L.orbx_search_window_batch_device(
            self._ctx, n, C.c_void_p(d_kps_un), kp_cap, C.c_void_p(d_cell_start), C.c_void_p(d_cell_idx), C.c_void_p(d_desc),
            C.c_void_p(d_u_right or None), C.c_void_p(d_taken or None), min_x, min_y, inv_w, inv_h, C.c_void_p(d_queries),
            q_cap, C.c_void_p(d_nq), C.c_void_p(d_best_idx), C.c_void_p(d_best_dist), int(sync)), self._ctx)

    def search_by_projection_batch_device(self, n, d_kps_un, kp_cap, d_cell_start, d_cell_idx, d_desc, d_u_right, d_taken,
                                          params, d_poses, d_queries, q_cap, d_nq, d_match, d_query_match, d_nmatches,
                                          sync=True):
        """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1365-1507) for a
        batch of frame pairs: `params` is a PROJECTION_PARAMS_DTYPE record (host), d_queries holds q_cap
        MAP_QUERY_DTYPE records per frame, d_poses 24 floats per frame (Tcw, Tlw); d_u_right / d_taken may be 0;
        see orbx_search_by_projection_batch_device."""
        P = np.ascontiguousarray(params, PROJECTION_PARAMS_DTYPE)
        _check(self._L.orbx_search_by_projection_batch_device(
            self._ctx, n, C.c_void_p(d_kps_un), kp_cap, C.c_void_p(d_cell_start), C.c_void_p(d_cell_idx), C.c_void_p(d_desc),
            C.c_void_p(d_u_right or None), C.c_void_p(d_taken or None), _ptr(P), C.c_void_p(d_poses), C.c_void_p(d_queries),
            q_cap, C.c_void_p(d_nq), C.c_void_p(d_match), C.c_void_p(d_query_match), C.c_void_p(d_nmatches), int(sync)),
            self._ctx)

    def search_local_map_batch_device(self, n, d_kps_un, kp_cap, d_cell_start, d_cell_idx, d_desc, d_u_right, d_taken, min_x,
                                      min_y, inv_w, inv_h, th, nn_ratio, d_queries, q_cap, d_nq, d_match, d_query_match,
                                      d_nmatches, sync=True):
        """ORBmatcher::SearchByProjection(F, vpMapPoints, th) (src/ORBmatcher.cc:45-129) for a batch of frames: d_queries
        holds q_cap LOCAL_QUERY_DTYPE records per frame; d_u_right / d_taken may be 0; see
        orbx_search_local_map_batch_device."""
        _check(self._L.orbx_search_local_map_batch_device(
            self._ctx, n, C.c_void_p(d_kps_un), kp_cap, C.c_void_p(d_cell_start), C.c_void_p(d_cell_idx), C.c_void_p(d_desc),
            C.c_void_p(d_u_right or None), C.c_void_p(d_taken or None), min_x, min_y, inv_w, inv_h, th, nn_ratio,
            C.c_void_p(d_queries), q_cap, C.c_void_p(d_nq), C.c_void_p(d_match), C.c_void_p(d_query_match),
            C.c_void_p(d_nmatches), int(sync)), self._ctx)

    def rgbd_depth_batch_device(self, n, d_depth, depth_is_f32, width, height, stride_bytes, frame_stride_bytes,
                                depth_map_factor, d_kps, d_kps_un, kp_cap, d_n, mbf, d_u_right, d_depth_out, sync=True):
        """Frame::ComputeStereoFromRGBD (src/Frame.cc:662-682) with the depth conversion of Tracking::GrabImageRGBD for a
        batch of frames; raw device addresses; see orbx_rgbd_depth_batch_device."""
        _check(self._L.orbx_rgbd_depth_batch_device(
            self._ctx, n, C.c_void_p(d_depth), int(depth_is_f32), width, height, stride_bytes, frame_stride_bytes,
            depth_map_factor, C.c_void_p(d_kps), C.c_void_p(d_kps_un), kp_cap, C.c_void_p(d_n), mbf, C.c_void_p(d_u_right),
            C.c_void_p(d_depth_out), int(sync)), self._ctx)

    def search_by_bow_batch_device(self, n, d_fv_node_kf, d_fv_idx_kf, d_fv_n_kf, d_desc_kf, d_kps_kf, d_valid_kf, kp_cap_kf,
                                   d_fv_node_f, d_fv_idx_f, d_fv_n_f, d_desc_f, d_kps_f, kp_cap_f, nn_ratio, check_orientation,
                                   d_match, d_nmatches, sync=True):
        """ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) (src/ORBmatcher.cc:159-290) for a batch of (keyframe, frame)
        pairs on the FeatureVectors bow_transform_batch_device leaves; raw device addresses; see
        orbx_search_by_bow_batch_device."""
        _check(self._L.orbx_search_by_bow_batch_device(
            self._ctx, n, C.c_void_p(d_fv_node_kf), C.c_void_p(d_fv_idx_kf), C.c_void_p(d_fv_n_kf), C.c_void_p(d_desc_kf),
            C.c_void_p(d_kps_kf), C.c_void_p(d_valid_kf), kp_cap_kf, C.c_void_p(d_fv_node_f), C.c_void_p(d_fv_idx_f),
            C.c_void_p(d_fv_n_f), C.c_void_p(d_desc_f), C.c_void_p(d_kps_f), kp_cap_f, nn_ratio, int(check_orientation),
            C.c_void_p(d_match), C.c_void_p(d_nmatches), int(sync)), self._ctx)

    def search_by_bow_keyframes_batch_device(self, n, d_fv_node_1, d_fv_idx_1, d_fv_n_1, d_desc_1, d_kps_1, d_valid_1, kp_cap_1,
                                             d_fv_node_2, d_fv_idx_2, d_fv_n_2, d_desc_2, d_kps_2, d_valid_2, kp_cap_2, nn_ratio,
                                             check_orientation, d_match12, d_nmatches, sync=True):
        """ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (src/ORBmatcher.cc:551-684) for a batch of keyframe pairs;
        see orbx_search_by_bow_keyframes_batch_device."""
        _check(self._L.orbx_search_by_bow_keyframes_batch_device(
            self._ctx, n, C.c_void_p(d_fv_node_1), C.c_void_p(d_fv_idx_1), C.c_void_p(d_fv_n_1), C.c_void_p(d_desc_1),
            C.c_void_p(d_kps_1), C.c_void_p(d_valid_1), kp_cap_1, C.c_void_p(d_fv_node_2), C.c_void_p(d_fv_idx_2),
            C.c_void_p(d_fv_n_2), C.c_void_p(d_desc_2), C.c_void_p(d_kps_2), C.c_void_p(d_valid_2), kp_cap_2, nn_ratio,
            int(check_orientation), C.c_void_p(d_match12), C.c_void_p(d_nmatches), int(sync)), self._ctx)

    def search_guided_batch_device(self, n, d_kps_un, kp_cap, d_cell_start, d_cell_idx, d_desc, d_taken, min_x, min_y, inv_w, inv_h,
                                   d_queries, q_cap, d_nq, level_up, max_dist, check_orientation, d_match, d_query_match,
                                   d_nmatches, sync=True):
        """The matching loops of ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist)
        (src/ORBmatcher.cc:1509-1636; level_up 1) and SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (:290-403; level_up 0,
        max_dist 50) on projected map points (GUIDED_QUERY_DTYPE records); see orbx_search_guided_batch_device."""
        _check(self._L.orbx_search_guided_batch_device(
            self._ctx, n, C.c_void_p(d_kps_un), kp_cap, C.c_void_p(d_cell_start), C.c_void_p(d_cell_idx), C.c_void_p(d_desc),
            C.c_void_p(d_taken), min_x, min_y, inv_w, inv_h, C.c_void_p(d_queries), q_cap, C.c_void_p(d_nq), int(level_up),
            int(max_dist), int(check_orientation), C.c_void_p(d_match), C.c_void_p(d_query_match), C.c_void_p(d_nmatches),
            int(sync)), self._ctx)

    def is_in_frustum_batch_device(self, n, params, d_poses, d_centres, d_points, q_cap, d_nq, d_queries_out, d_n_in_view,
                                   sync=True):
        """Frame::isInFrustum with MapPoint::PredictScale (src/Frame.cc:271-327, src/MapPoint.cc:402-417) for a batch of
        frames and their local map points (FRUSTUM_POINT_DTYPE records in, LOCAL_QUERY_DTYPE records out); `params` is a
        FrustumParams; see orbx_is_in_frustum_batch_device."""
        _check(self._L.orbx_is_in_frustum_batch_device(
            self._ctx, n, C.byref(params), C.c_void_p(d_poses), C.c_void_p(d_centres), C.c_void_p(d_points), q_cap,
            C.c_void_p(d_nq), C.c_void_p(d_queries_out), C.c_void_p(d_n_in_view), int(sync)), self._ctx)

    def fuse_project_batch_device(self, n, params, d_poses, d_centres, d_points, q_cap, d_nq, d_queries_out, d_n_projected,
                                  sync=True):
        """The tests in front of the search of ORBmatcher::Fuse(pKF, vpMapPoints, th) (src/ORBmatcher.cc:881-927) for a batch
        of keyframes (FRUSTUM_POINT_DTYPE records in, WINDOW_QUERY_DTYPE records out, radius -1 for a dropped point); see
        orbx_fuse_project_batch_device."""
        _check(self._L.orbx_fuse_project_batch_device(
            self._ctx, n, C.byref(params), C.c_void_p(d_poses), C.c_void_p(d_centres), C.c_void_p(d_points), q_cap,
            C.c_void_p(d_nq), C.c_void_p(d_queries_out), C.c_void_p(d_n_projected), int(sync)), self._ctx)

    def project_guided_batch_device(self, n, params, d_poses, d_centres, d_points, d_angles, q_cap, d_nq, relocalisation,
                                    d_queries_out, d_n_projected, sync=True):
        """The projections in front of the guided searches (src/ORBmatcher.cc:321-360, or :1534-1563 with
        relocalisation=True) for a batch of frames / keyframes: FRUSTUM_POINT_DTYPE records in, GUIDED_QUERY_DTYPE records
        out; d_angles may be 0.  See orbx_project_guided_batch_device."""
        _check(self._L.orbx_project_guided_batch_device(
            self._ctx, n, C.byref(params), C.c_void_p(d_poses), C.c_void_p(d_centres), C.c_void_p(d_points),
            C.c_void_p(d_angles), q_cap, C.c_void_p(d_nq), int(relocalisation), C.c_void_p(d_queries_out),
            C.c_void_p(d_n_projected), int(sync)), self._ctx)

    def sim3_project_batch_device(self, n, params, d_poses, d_points, q_cap, d_nq, d_queries_out, d_n_projected, sync=True):
        """The projection of one direction of ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1195-1226) for a batch of keyframe
        pairs: d_poses holds [R1w | t1w] and [sR21 | t21] per pair (24 floats), FRUSTUM_POINT_DTYPE records in,
        WINDOW_QUERY_DTYPE records out.  See orbx_sim3_project_batch_device."""
        _check(self._L.orbx_sim3_project_batch_device(
            self._ctx, n, C.byref(params), C.c_void_p(d_poses), C.c_void_p(d_points), q_cap, C.c_void_p(d_nq),
            C.c_void_p(d_queries_out), C.c_void_p(d_n_projected), int(sync)), self._ctx)

    def fuse_search_batch_device(self, n, d_kps_un, kp_cap, d_cell_start, d_cell_idx, d_desc, d_u_right, min_x, min_y, inv_w,
                                 inv_h, d_queries, q_cap, d_nq, d_best_idx, d_best_dist, sync=True):
        """The search of ORBmatcher::Fuse(pKF, vpMapPoints, th) (src/ORBmatcher.cc:929-986) for a batch of keyframes and
        their projected map points (WINDOW_QUERY_DTYPE records); see orbx_fuse_search_batch_device."""
        _check(self._L.orbx_fuse_search_batch_device(
            self._ctx, n, C.c_void_p(d_kps_un), kp_cap, C.c_void_p(d_cell_start), C.c_void_p(d_cell_idx), C.c_void_p(d_desc),
            C.c_void_p(d_u_right), min_x, min_y, inv_w, inv_h, C.c_void_p(d_queries), q_cap, C.c_void_p(d_nq),
            C.c_void_p(d_best_idx), C.c_void_p(d_best_dist), int(sync)), self._ctx)

    def search_for_triangulation_batch_device(self, n, d_fv_node_1, d_fv_idx_1, d_fv_n_1, d_desc_1, d_kps_un_1, d_has_mp_1,
                                              d_u_right_1, kp_cap_1, d_fv_node_2, d_fv_idx_2, d_fv_n_2, d_desc_2, d_kps_un_2,
                                              d_has_mp_2, d_u_right_2, kp_cap_2, d_f12, d_epipole, only_stereo,
                                              check_orientation, d_match12, d_nmatches, sync=True):
        """ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (src/ORBmatcher.cc:686-852) for a
        batch of keyframe pairs; d_u_right_* may be 0 (no stereo measurements).  See
        orbx_search_for_triangulation_batch_device."""
        _check(self._L.orbx_search_for_triangulation_batch_device(
            self._ctx, n, C.c_void_p(d_fv_node_1), C.c_void_p(d_fv_idx_1), C.c_void_p(d_fv_n_1), C.c_void_p(d_desc_1),
            C.c_void_p(d_kps_un_1), C.c_void_p(d_has_mp_1), C.c_void_p(d_u_right_1), kp_cap_1, C.c_void_p(d_fv_node_2),
            C.c_void_p(d_fv_idx_2), C.c_void_p(d_fv_n_2), C.c_void_p(d_desc_2), C.c_void_p(d_kps_un_2), C.c_void_p(d_has_mp_2),
            C.c_void_p(d_u_right_2), kp_cap_2, C.c_void_p(d_f12), C.c_void_p(d_epipole), int(only_stereo),
            int(check_orientation), C.c_void_p(d_match12), C.c_void_p(d_nmatches), int(sync)), self._ctx)

    def search_for_initialization_batch_device(self, n, d_kps_un_1, d_desc_1, d_n_1, kp_cap_1, d_prev_matched, d_kps_un_2,
                                               d_cell_start_2, d_cell_idx_2, d_desc_2, kp_cap_2, min_x, min_y, inv_w, inv_h,
                                               window_size, nn_ratio, check_orientation, d_match12, d_nmatches, sync=True):
        """ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (src/ORBmatcher.cc:418-549)
        for a batch of frame pairs; d_prev_matched is read and updated.  See orbx_search_for_initialization_batch_device."""
        _check(self._L.orbx_search_for_initialization_batch_device(
            self._ctx, n, C.c_void_p(d_kps_un_1), C.c_void_p(d_desc_1), C.c_void_p(d_n_1), kp_cap_1, C.c_void_p(d_prev_matched),
            C.c_void_p(d_kps_un_2), C.c_void_p(d_cell_start_2), C.c_void_p(d_cell_idx_2), C.c_void_p(d_desc_2), kp_cap_2,
            min_x, min_y, inv_w, inv_h, int(window_size), nn_ratio, int(check_orientation), C.c_void_p(d_match12),
            C.c_void_p(d_nmatches), int(sync)), self._ctx)

    # ---- the stage after the extractor: Frame::ComputeStereoMatches (src/Frame.cc:485-659) ----
    def stereo_match(self, right, mb, mbf, img_left=0, img_right=0):
        """Associates image `img_left` of this extractor's last host-memory extraction with image
        `img_right` of `right`'s (another ORBextractor, or this one).  Returns (mvuRight, mvDepth):
        float32 arrays with one value per left keypoint, -1 where there is no match."""
        cap = self.max_keypoints(*self._last_size())
        u = np.empty(cap, np.float32)
        d = np.empty(cap, np.float32)
        n = C.c_int(0)
        _check(self._L.orbx_stereo_match(self._ctx, img_left, right._ctx, img_right, mb, mbf, _ptr(u), _ptr(d), cap,
                                         C.byref(n)), self._ctx)
        return u[:n.value].copy(), d[:n.value].copy()

    def stereo_match_batch_device(self, n_pairs, d_kps, kp_cap, d_desc, d_n, mb, mbf, d_u_right, d_depth, sync=True):
        """Pairs (2p, 2p+1) of the last extract_batch_device; raw device addresses (ints); see
        orbx_stereo_match_batch_device."""
        _check(self._L.orbx_stereo_match_batch_device(self._ctx, n_pairs, C.c_void_p(d_kps), kp_cap, C.c_void_p(d_desc),
                                                      C.c_void_p(d_n), mb, mbf, C.c_void_p(d_u_right),
                                                      C.c_void_p(d_depth), int(sync)), self._ctx)

    def stereo_match_batch_device_noalloc(self, n_pairs, d_kps, kp_cap, d_desc, d_n, mb, mbf, d_u_right, d_depth,
                                          sync=True):
        """stereo_match_batch_device with the kernels' scratch in the library's fixed array: allocates
        nothing; sizes beyond NOALLOC_SCRATCH_BYTES raise; see orbx_stereo_match_batch_device_noalloc."""
        _check(self._L.orbx_stereo_match_batch_device_noalloc(self._ctx, n_pairs, C.c_void_p(d_kps), kp_cap,
                                                              C.c_void_p(d_desc), C.c_void_p(d_n), mb, mbf,
                                                              C.c_void_p(d_u_right), C.c_void_p(d_depth), int(sync)),
               self._ctx)

    # ---- pyramid read-back: the public member mvImagePyramid (include/ORBextractor.h:85) ----
    def pyramid_level(self, img_idx, level, padded=False):
        """Level image of the last extraction.  padded=True returns the view the reference
        keeps: the level inside a buffer with a 19-px reflect-101 border
        (src/ORBextractor.cc:1227-1245); the returned array is the un-padded ROI of it."""
        g = host_level_geometry(self.nfeatures, self.scaleFactor, self.nlevels, *self._last_size(), level)
        w, h = g.width, g.height
        E = EDGE_THRESHOLD
        if padded:
            buf = np.empty((h + 2 * E, w + 2 * E), np.uint8)
            roi = buf[E:E + h, E:E + w]
        else:
            buf = roi = np.empty((h, w), np.uint8)
        ww, hh = C.c_int(), C.c_int()
        _check(self._L.orbx_pyramid_level(self._ctx, img_idx, level, roi.ctypes.data_as(C.c_void_p), roi.strides[0],
                                          C.byref(ww), C.byref(hh)), self._ctx)
        assert (ww.value, hh.value) == (w, h)
        if padded:
            buf[:] = np.pad(roi, E, mode="reflect")
            return buf[E:E + h, E:E + w]
        return roi

    def _last_size(self):
        return self._size

    # ---- test / measurement hooks ----
    def debug_read(self, what, img_idx, level, max_bytes=1 << 26):
        buf = np.zeros(max_bytes, np.uint8)
        n = C.c_size_t(0)
        _check(self._L.orbx_debug_read(self._ctx, what, img_idx, level, _ptr(buf), max_bytes, C.byref(n)), self._ctx)
        return buf[:n.value].copy()

    def set_debug_flag(self, flag, value):
        """One of the FLAG_* diagnostic switches of a live context (see orbx_set_debug_flag)."""
        _check(self._L.orbx_set_debug_flag(self._ctx, int(flag), int(value)), self._ctx)

    def set_profiling(self, on):
        _check(self._L.orbx_set_profiling(self._ctx, int(on)), self._ctx)

    def stage_ms(self):
        ms = np.zeros(NUM_STAGES, np.float32)
        _check(self._L.orbx_get_stage_ms(self._ctx, _ptr(ms)), self._ctx)
        return {self._L.orbx_stage_name(i).decode(): float(ms[i]) for i in range(NUM_STAGES)}


FRUSTUM_POINT_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("z", np.float32), ("nx", np.float32),
                                ("ny", np.float32), ("nz", np.float32), ("min_dist", np.float32), ("max_dist", np.float32),
                                ("flags", np.int32), ("desc", np.uint8, (32,))])
assert FRUSTUM_POINT_DTYPE.itemsize == 68


class FrustumParams(C.Structure):
    """orbx_frustum_params."""
    _fields_ = [(k, C.c_float) for k in ("fx", "fy", "cx", "cy", "mbf", "min_x", "max_x", "min_y", "max_y",
                                         "view_cos_limit", "log_scale_factor", "th")]


GUIDED_QUERY_DTYPE = np.dtype([("u", np.float32), ("v", np.float32), ("angle", np.float32), ("level", np.int32),
                               ("radius", np.float32), ("flags", np.int32), ("desc", np.uint8, (32,))])
assert GUIDED_QUERY_DTYPE.itemsize == 56
WINDOW_QUERY_DTYPE = np.dtype([("u", np.float32), ("v", np.float32), ("radius", np.float32), ("min_level", np.int32),
                               ("max_level", np.int32), ("ur", np.float32), ("desc", np.uint8, (32,))])
assert WINDOW_QUERY_DTYPE.itemsize == 56
MAP_QUERY_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("z", np.float32), ("octave", np.int32),
                            ("angle", np.float32), ("flags", np.int32), ("desc", np.uint8, (32,))])
assert MAP_QUERY_DTYPE.itemsize == 56
LOCAL_QUERY_DTYPE = np.dtype([("u", np.float32), ("v", np.float32), ("ur", np.float32), ("level", np.int32),
                              ("view_cos", np.float32), ("flags", np.int32), ("desc", np.uint8, (32,))])
assert LOCAL_QUERY_DTYPE.itemsize == 56
PROJECTION_PARAMS_DTYPE = np.dtype([(k, np.float32) for k in ("fx", "fy", "cx", "cy", "mbf", "mb", "min_x", "min_y", "max_x",
                                                              "max_y", "inv_w", "inv_h", "th")] +
                                   [("mono", np.int32), ("check_orientation", np.int32)])
assert PROJECTION_PARAMS_DTYPE.itemsize == 60


class ORBVocabulary:
    """The vocabulary tree of ORB_SLAM2::ORBVocabulary (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>,
    reference include/ORBVocabulary.h) for ORBextractor.bow_transform_batch_device: built from per-node arrays
    (entry 0 = root) or read from the text file ORB_SLAM2 ships (`ORBVocabulary.load_text`)."""

    def __init__(self, k, L, scoring, weighting, parent, is_leaf, desc, weight, _handle=None):
        self._L = load_library()
        if _handle is not None:
            self._v = _handle
            return
        parent = np.ascontiguousarray(parent, np.int32)
        is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        weight = np.ascontiguousarray(weight, np.float64)
        n = len(parent)
        if not (len(is_leaf) == n and len(desc) == n and len(weight) == n):
            raise OrbxError("vocabulary arrays differ in length")
        v = C.c_void_p()
        _check(self._L.orbx_voc_create(k, L, scoring, weighting, n, _ptr(parent), _ptr(is_leaf), _ptr(desc), _ptr(weight),
                                       C.byref(v)))
        self._v = v

    @classmethod
    def load_text(cls, path):
        L = load_library()
        v = C.c_void_p()
        _check(L.orbx_voc_load_text(str(path).encode(), C.byref(v)))
        return cls(0, 0, 0, 0, None, None, None, None, _handle=v)

    def info(self):
        k, L, nn, nw = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        _check(self._L.orbx_voc_info(self._v, C.byref(k), C.byref(L), C.byref(nn), C.byref(nw)))
        return {"k": k.value, "L": L.value, "n_nodes": nn.value, "n_words": nw.value}

    def node(self, i):
        """(n_children, first_child, centre[32], weight, word id) of node i."""
        nc, fc, w, wd = C.c_int(0), C.c_uint32(0), C.c_double(0), C.c_uint32(0)
        d = np.zeros(32, np.uint8)
        _check(self._L.orbx_voc_node(self._v, i, C.byref(nc), C.byref(fc), _ptr(d), C.byref(w), C.byref(wd)))
        return nc.value, fc.value, d, w.value, wd.value

    def close(self):
        if getattr(self, "_v", None):
            self._L.orbx_voc_destroy(self._v)
            self._v = None

    def __del__(self):
        self.close()
