"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them bit for bit."""
import glob
import hashlib
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _same(kps, desc, g):
    assert len(kps) == len(g["keypoints"])
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert (kps[f] == g["keypoints"][f]).all(), f
    assert (desc == g["descriptors"]).all()


def test_fixtures_present():
    assert len(FILES) >= 2


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path, allow_pickle=False)
    nl = int(g["nlevels"])
    orc = oracle.Oracle(int(g["nfeatures"]), float(g["scale_factor"]), nl, int(g["ini_th"]), int(g["min_th"]))
    kps, desc = orc.extract(g["image"])
    _same(kps, desc, g)
    assert [_sha(orc.level(l)) for l in range(nl)] == list(g["level_sha256"])
    assert [_sha(orc.blurred(l)) for l in range(nl)] == list(g["blurred_sha256"])
    assert [len(orc.candidates(l)) for l in range(nl)] == list(g["candidates_per_level"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_reproduces_golden(path):
    import orb_slam2_amd as oa
    g = np.load(path, allow_pickle=False)
    nl = int(g["nlevels"])
    img = g["image"]
    ex = oa.ORBextractor(int(g["nfeatures"]), float(g["scale_factor"]), nl, int(g["ini_th"]), int(g["min_th"]),
                         max_width=img.shape[1], max_height=img.shape[0])
    kps, desc = ex(img)
    _same(kps, desc, g)
    for l in range(nl):
        assert _sha(ex.pyramid_level(0, l)) == g["level_sha256"][l]
        w, h = ex.pyramid_level(0, l).shape[::-1]
        assert _sha(ex.debug_read(oa.DBG_BLURRED, 0, l).reshape(h, w)) == g["blurred_sha256"][l]
        assert len(ex.debug_read(oa.DBG_CANDIDATES, 0, l)) // 12 == g["candidates_per_level"][l]
        assert len(ex.debug_read(oa.DBG_LEVEL_KPS, 0, l)) // 12 == g["keypoints_per_level"][l]
    ex.close()
