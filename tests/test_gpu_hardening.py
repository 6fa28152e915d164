"""Device-side counts and coordinates are not trusted by the kernels that consume them: a value outside the
array it indexes is cut to the bound and the batch ends in ORBX_E_DEVICE_DATA instead of a GPU fault.  A
test hook of the library (ORBX_FLAG_POISON) overwrites what a producer left in device memory."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("what,name", [(1, "count of a FAST cell"), (2, "count of a level's retained points"),
                                       (3, "coordinates of a candidate point")])
def test_poisoned_device_data_is_reported_not_faulted(oracle, what, name):
    w, h, nf, nl = 640, 480, 1000, 8
    imgs = [synth.make_frame(w, h, 1, 40 + i) for i in range(3)]
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=3, keep_pyramid=False)
    orc = oracle.Oracle(nf, 1.2, nl, 20, 7)
    exp = [orc.extract(im) for im in imgs]

    def check_clean():
        for (kps, desc), (okps, odesc) in zip(ex.extract_batch(imgs), exp):
            assert len(kps) == len(okps)
            for f in oa.KP_DTYPE.names:
                assert (kps[f] == okps[f]).all(), f
            assert (desc == odesc).all()

    check_clean()
    ex.set_debug_flag(oa.FLAG_POISON, what)
    for attempt in range(2):     # the error is reported per batch, and the context stays usable
        with pytest.raises(oa.OrbxError) as e:
            ex.extract_batch(imgs)
        assert "(%d)" % oa.E_DEVICE_DATA in str(e.value), (name, str(e.value))
    # the single-frame path (one captured graph launch) reports it as well
    with pytest.raises(oa.OrbxError) as e:
        ex(imgs[0])
    assert "(%d)" % oa.E_DEVICE_DATA in str(e.value), (name, str(e.value))
    ex.set_debug_flag(oa.FLAG_POISON, 0)
    check_clean()                # and afterwards the context extracts as before
    kps, desc = ex(imgs[0])
    assert len(kps) == len(exp[0][0]) and (desc == exp[0][1]).all()
    ex.close()
