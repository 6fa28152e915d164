"""The drop-in C++ class ORB_SLAM2::ORBextractor (include/ORBextractor.h) on the GPU, through a
small harness compiled against a test-only OpenCV stand-in, against the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

from orb_slam2_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "orbextractor_harness")

KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
               ("octave", "<i4"), ("class_id", "<i4")])


def test_cpp_class_matches_oracle(oracle, tmp_path):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    w, h, stride, nf, nl = 752, 480, 800, 1200, 8
    img = synth.make_frame(w, h, 4, 3)
    buf = np.zeros((h, stride), np.uint8)
    buf[:, :w] = img
    raw, out = tmp_path / "in.raw", tmp_path / "out.bin"
    buf.tofile(raw)
    subprocess.check_call([EXE, "extract", str(nf), "1.2", str(nl), "20", "7", str(w), str(h), str(stride),
                           str(raw), str(out)], timeout=120)
    blob = out.read_bytes()
    k = int(np.frombuffer(blob, np.int32, 1)[0])
    kps = np.frombuffer(blob, KP, k, 4)
    desc = np.frombuffer(blob, np.uint8, k * 32, 4 + 28 * k).reshape(k, 32)
    orc = oracle.Oracle(nf, 1.2, nl, 20, 7)
    okps, odesc = orc.extract(img)
    assert k == len(okps)
    for f in KP.names:
        assert (kps[f] == okps[f]).all(), f
    assert (desc == odesc).all()
    off = 4 + 60 * k
    for l in range(nl):
        lw, lh = (int(v) for v in np.frombuffer(blob, np.int32, 2, off))
        off += 8
        assert (lw, lh) == orc.level_dims(l)
        padded = np.frombuffer(blob, np.uint8, (lw + 38) * (lh + 38), off).reshape(lh + 38, lw + 38)
        off += padded.size
        assert (padded == orc.padded_level(l)).all(), "mvImagePyramid level %d" % l
    assert off == len(blob)
