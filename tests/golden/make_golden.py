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


if __name__ == "__main__":
    make("synth_400x300_f600_l6", synth.make_frame(400, 300, 11, 0), 600, 6)
    ref_dir = "/root/reference/Examples/Monocular/rgbd_dataset_freiburg1_xyz/rgb"
    if os.path.isdir(ref_dir):
        from PIL import Image
        f = sorted(os.listdir(ref_dir))[0]
        gray = np.asarray(Image.open(os.path.join(ref_dir, f)).convert("L"), dtype=np.uint8)
        make("tum_fr1_xyz_" + f.replace(".png", "").replace(".", "_") + "_f1000_l8", gray, 1000, 8)
