"""The in-process multi-GPU host path (orbx_pool_*) rehearsed on ONE device: a pool of two, then three workers that
all use device 0, a stream of host frames cut by frame, every frame's result at its own position and equal to the
oracle's."""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("workers,n_frames,max_batch", [(2, 44, 16), (3, 37, 6), (2, 3, 4)])
def test_pool_stream_in_frame_order(oracle, workers, n_frames, max_batch):
    w, h, nf, nl = 752, 480, 1200, 8
    base = synth.make_batch(6, w + 16, h, 4, 800)
    frames = [np.ascontiguousarray(base[i % 6][:, (i * 3) % 16:(i * 3) % 16 + w]) for i in range(n_frames)]
    pool = oa.ExtractorPool(nf, 1.2, nl, 20, 7, [0] * workers, max_width=w, max_height=h, max_batch=max_batch)
    assert pool.workers() == workers
    kps, desc, cnt = pool.extract_stream(frames)
    kps2, desc2, cnt2 = pool.extract_stream(frames)          # the workers and their contexts are reused
    assert (cnt == cnt2).all() and (desc == desc2).all()
    orc = oracle.Oracle(nf, 1.2, nl, 20, 7)
    cache = {}
    for i, f in enumerate(frames):
        key = (i % 6, (i * 3) % 16)
        if key not in cache:
            cache[key] = orc.extract(f)
        okps, odesc = cache[key]
        assert cnt[i] == len(okps), i
        for name in oa.KP_DTYPE.names:
            assert (kps[i, :cnt[i]][name] == okps[name]).all(), (i, name)
        assert (desc[i, :cnt[i]] == odesc).all(), i
    pool.close()


def test_pool_arguments():
    with pytest.raises(oa.OrbxError):
        oa.ExtractorPool(500, 1.2, 4, 20, 7, [], max_width=320, max_height=240)
    with pytest.raises(oa.OrbxError):
        oa.ExtractorPool(500, 1.2, 4, 20, 7, [99], max_width=320, max_height=240)      # no such device
    with pytest.raises(oa.OrbxError):
        oa.ExtractorPool(500, 1.2, 4, 20, 7, [0], max_width=320, max_height=240, max_batch=1)
