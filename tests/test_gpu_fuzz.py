"""A short run of the randomised parity sweep (tools/fuzz_parity.py) as a test: random image sizes, extractor
parameters and image kinds, single-frame and batch entry points, every stage compared bit-exactly with the oracle.
The long sweeps are run with the tool itself."""
import os
import sys

import numpy as np
import pytest

import orb_slam2_amd as oa

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fuzz_parity  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,batch", [(101, 8), (102, 17)])
def test_random_configurations(oracle, seed, batch):
    rng = np.random.default_rng(seed)
    done = 0
    for i in range(30):
        c = fuzz_parity.random_case(rng, 1400, 1000)
        ex = oa.ORBextractor(c["nf"], c["scale"], c["nlev"], c["ini"], c["mn"], max_width=c["w"], max_height=c["h"],
                             max_batch=batch)
        orc = oracle.Oracle(c["nf"], c["scale"], c["nlev"], c["ini"], c["mn"])
        img = fuzz_parity.make_image(c)
        okps, odesc = orc.extract(img)
        kps, desc = ex(img)
        assert fuzz_parity.compare(ex, orc, 0, kps, desc, okps, odesc, c["nlev"]) is None, c
        if i % 5 == 0:
            imgs = [fuzz_parity.make_image(c, k + 1) for k in range(batch - 1)] + [img]
            bk, bd = ex.extract_batch(imgs)[-1]
            assert fuzz_parity.compare(ex, orc, batch - 1, bk, bd, okps, odesc, c["nlev"], lists=batch < 16) is None, c
        ex.close()
        done += 1
    assert done == 30
