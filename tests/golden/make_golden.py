#!/usr/bin/env python3
"""Generate the committed golden vectors from the CPU oracle.

Run in the development container:  python tests/golden/make_golden.py
Inputs: one synthetic frame (generator in orb_slam2_amd/synth.py) and one real frame of the TUM
fr1/xyz sequence that ships with the reference (Examples/Monocular/rgbd_dataset_freiburg1_xyz/rgb),
converted to 8-bit gray with PIL.  The image is data, stored compressed inside the .npz; the
expected keypoints, descriptors and per-level digests are outputs of oracle/liborb_oracle.so.
The reference itself cannot run here (it needs OpenCV), so these vectors pin the oracle and the
HIP path against each other and against drift -- not against the reference ("parity unpinned").
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make(name, img, nfeatures, nlevels):
    orc = oracle_lib.Oracle(nfeatures, 1.2, nlevels, 20, 7)
    kps, desc = orc.extract(img)
    levels = [digest(orc.level(l)) for l in range(nlevels)]
    blurred = [digest(orc.blurred(l)) for l in range(nlevels)]
    ncand = [len(orc.candidates(l)) for l in range(nlevels)]
    nkp = [orc.level_keypoint_count(l) for l in range(nlevels)]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img, nfeatures=nfeatures, nlevels=nlevels,
                        scale_factor=np.float32(1.2), ini_th=20, min_th=7, keypoints=kps, descriptors=desc,
                        level_sha256=np.array(levels), blurred_sha256=np.array(blurred),
                        candidates_per_level=np.array(ncand), keypoints_per_level=np.array(nkp))
    print(name, img.shape, "K =", len(kps), "cands", ncand)


TUM1_K = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)      # Examples/RGB-D/TUM1.yaml
TUM1_DIST = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)
TUM1_BF, TUM1_DEPTH_FACTOR = 40.0, 5000.0


def frame_of(img, depth):
    """What Frame::Frame(RGB-D) derives from an image and its depth map (src/Frame.cc:121-176): keypoints,
    descriptors, undistorted keypoints, the grid, and mvuRight / mvDepth of ComputeStereoFromRGBD (:661-682)."""
    orc = oracle_lib.Oracle(1000, 1.2, 8, 20, 7)
    kps, desc = orc.extract(img)
    xy = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
    un = oracle_lib.undistort_points(xy, TUM1_K, TUM1_DIST)
    h, w = img.shape
    c = oracle_lib.undistort_points(np.array([[0, 0], [w, 0], [0, h], [w, h]], np.float32), TUM1_K, TUM1_DIST)
    min_x, max_x = min(c[0, 0], c[2, 0]), max(c[1, 0], c[3, 0])
    min_y, max_y = min(c[0, 1], c[1, 1]), max(c[2, 1], c[3, 1])
    inv_w, inv_h = np.float32(64) / np.float32(max_x - min_x), np.float32(48) / np.float32(max_y - min_y)
    start, idx = oracle_lib.assign_grid(un, min_x, min_y, inv_w, inv_h)
    # the samples times Tracking's mDepthMapFactor = 1.0f / 5000 (src/Tracking.cc:145-149, 230-231)
    u_right, d = oracle_lib.stereo_from_rgbd(depth, np.float32(1.0) / np.float32(TUM1_DEPTH_FACTOR), xy, un[:, 0], TUM1_BF)
    return dict(kps=kps, desc=desc, un=un, start=start, idx=idx, depth=d, u_right=u_right,
                bounds=np.array([min_x, min_y, max_x, max_y, inv_w, inv_h], np.float32),
                scale=np.array(orc.params.scale[:8], np.float32))


def make_projection(name, img0, depth0, img1, depth1):
    """Two consecutive RGB-D frames of the sequence as (last, current): the last frame's keypoints with a depth
    become map points (Frame::UnprojectStereo, src/Frame.cc:684-697, world = the last camera), the current
    frame is searched with the pose of the last one -- Tracking::TrackWithMotionModel before any motion is
    known (src/Tracking.cc:1029), th = 15 as for RGB-D input."""
    last, cur = frame_of(img0, depth0), frame_of(img1, depth1)
    n0 = len(last["kps"])
    q = np.zeros(n0, oracle_lib.MAP_QUERY_DTYPE)
    z = last["depth"]
    ok = z > 0
    q["x"] = np.where(ok, (last["un"][:, 0] - TUM1_K[2]) * z / TUM1_K[0], 0)
    q["y"] = np.where(ok, (last["un"][:, 1] - TUM1_K[3]) * z / TUM1_K[1], 0)
    q["z"] = np.where(ok, z, 1)
    q["octave"], q["angle"], q["desc"] = last["kps"]["octave"], last["kps"]["angle"], last["desc"]
    q["flags"] = np.where(ok, 1, 2)
    P = np.zeros((), oracle_lib.PROJ_PARAMS_DTYPE)
    P["fx"], P["fy"], P["cx"], P["cy"] = TUM1_K
    P["mbf"], P["mb"] = TUM1_BF, np.float32(TUM1_BF) / TUM1_K[0]
    P["min_x"], P["min_y"], P["max_x"], P["max_y"], P["inv_w"], P["inv_h"] = cur["bounds"]
    P["th"], P["mono"], P["check_orientation"] = 15.0, 0, 1
    eye = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32)
    n, match, qmatch = oracle_lib.search_by_projection(cur["un"], cur["kps"]["octave"].astype(np.int32), cur["kps"]["angle"],
                                                       cur["desc"], cur["u_right"], None, cur["start"], cur["idx"], cur["scale"],
                                                       P, eye, eye, q)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img1, queries=q.view(np.uint8).reshape(n0, 56),
                        u_right=cur["u_right"], params=np.frombuffer(P.tobytes(), np.uint8), K=TUM1_K, dist=TUM1_DIST,
                        nmatches=n, match=match, query_match=qmatch)
    print(name, "map points", int(ok.sum()), "of", n0, "features", len(cur["kps"]), "nmatches", n)


def make_rgbd(name, img, depth):
    """One RGB-D frame of the sequence: the raw 16-bit depth map and what Frame::ComputeStereoFromRGBD
    (src/Frame.cc:662-682) makes of it for the frame's keypoints, with the TUM1 settings (DepthMapFactor 5000, so
    Tracking multiplies the samples by 1.0f / 5000.0f; Camera.bf 40)."""
    orc = oracle_lib.Oracle(1000, 1.2, 8, 20, 7)
    kps, _ = orc.extract(img)
    xy = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
    un = oracle_lib.undistort_points(xy, TUM1_K, TUM1_DIST)
    factor = np.float32(1.0) / np.float32(TUM1_DEPTH_FACTOR)
    ur, z = oracle_lib.stereo_from_rgbd(depth, factor, xy, un[:, 0], TUM1_BF)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img, depth=depth, K=TUM1_K, dist=TUM1_DIST,
                        factor=factor, mbf=np.float32(TUM1_BF), u_right=ur, depth_out=z, n=len(kps))
    print(name, "K =", len(kps), "with depth:", int((z > 0).sum()))


if __name__ == "__main__":
    only = sys.argv[1] if len(sys.argv) > 1 else ""   # "extractor", "projection" or "rgbd": regenerate one family only
    if only in ("", "extractor"):
        make("synth_400x300_f600_l6", synth.make_frame(400, 300, 11, 0), 600, 6)
    ref_dir = "/root/reference/Examples/Monocular/rgbd_dataset_freiburg1_xyz/rgb"
    if os.path.isdir(ref_dir):
        from PIL import Image
        if only in ("", "extractor"):
            f = sorted(os.listdir(ref_dir))[0]
            gray = np.asarray(Image.open(os.path.join(ref_dir, f)).convert("L"), dtype=np.uint8)
            make("tum_fr1_xyz_" + f.replace(".png", "").replace(".", "_") + "_f1000_l8", gray, 1000, 8)
        if only in ("", "rgbd"):
            seq = os.path.dirname(ref_dir)
            r = open("/root/reference/Examples/RGB-D/associations/fr1_xyz.txt").readline().split()
            make_rgbd("rgbd_tum_fr1_xyz_frame_0", np.asarray(Image.open(os.path.join(seq, r[1])).convert("L"), dtype=np.uint8),
                      np.asarray(Image.open(os.path.join(seq, r[3])), dtype=np.uint16))
        if only not in ("", "projection"):
            sys.exit(0)
        # the first two associated (rgb, depth) pairs of Examples/RGB-D/associations/fr1_xyz.txt
        seq = os.path.dirname(ref_dir)
        rows = [l.split() for l in open("/root/reference/Examples/RGB-D/associations/fr1_xyz.txt")][:2]
        g = [np.asarray(Image.open(os.path.join(seq, r[1])).convert("L"), dtype=np.uint8) for r in rows]
        d = [np.asarray(Image.open(os.path.join(seq, r[3])), dtype=np.uint16) for r in rows]
        make_projection("projection_tum_fr1_xyz_frames_0_1", g[0], d[0], g[1], d[1])
