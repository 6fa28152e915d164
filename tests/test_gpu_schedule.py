"""GPU ordering tests of the launch schedule: random sequences of calls on one context, enqueued without waiting,
must give the results of the same calls made one at a time.

A batch of resident images builds its pyramid and blur ahead of the context's main stream; a batch behind a copy, a
staging kernel, a read-back or the stereo stage is ordered behind them.  The sequences below mix all of these.
"""
import numpy as np
import pytest

import orb_slam2_amd as oa
from orb_slam2_amd import synth

pytestmark = pytest.mark.gpu

# batches long enough on the GPU (a few hundred microseconds) that the host runs ahead and work queues up on the streams
W, H, NF, NL, N = 640, 480, 1000, 8, 96


class _Seq:
    """One random sequence of operations and the buffers it needs."""

    def __init__(self, ex, seed, length):
        from hip_rt import DeviceArray
        self.ex, self.DA = ex, DeviceArray
        self.cap = ex.max_keypoints(W, H)
        rng = np.random.default_rng(seed)
        self.inputs = [DeviceArray.from_host(np.tile(synth.make_batch(8, W, H, 4, 7000 + 31 * seed + i), (N // 8, 1, 1))) for i in range(4)]
        odd = np.zeros((N, H, W + 3), np.uint8)                       # stride 323: the copy path of the device call
        odd[:, :, :W] = np.tile(synth.make_batch(8, W, H, 4, 7100 + seed), (N // 8, 1, 1))
        self.odd = DeviceArray.from_host(odd)
        # two colour sources in turn: every conversion changes the grey buffer, so a batch that read it too early
        # describes the other source (or the zeros it started with)
        self.rgb = [DeviceArray.from_host(np.repeat(np.tile(synth.make_batch(8, W, H, 4, 7200 + 50 * j + seed), (N // 8, 1, 1))[..., None], 3, axis=3))
                    for j in range(2)]
        self.conv = DeviceArray((N, H, W), np.uint8, fill_byte=0)
        self.ops = [str(rng.choice(["batch", "batch", "batch", "odd", "gray", "stereo", "level", "sync"])) for _ in range(length)]
        self.ops[0] = "batch"
        self.ops[3] = self.ops[9] = "gray"
        self.src = [int(rng.integers(0, 4)) for _ in range(length)]
        self.frames = [int(rng.choice([N, N, N // 2, 8])) for _ in range(length)]   # batches of different sizes in one sequence

    def run(self, wait):
        """Returns the list of everything the sequence produced, in order."""
        DA, ex, cap = self.DA, self.ex, self.cap
        outs, res, keep = [], [], []
        self.conv.fill(0)
        n_gray = 0
        last_n = N
        # every buffer is allocated and cleared before the first call: clearing one synchronises the device
        pool = [(DA((N, cap, 28), np.uint8, fill_byte=0), DA((N, cap, 32), np.uint8, fill_byte=0), DA((N,), np.int32, fill_byte=0))
                for op in self.ops if op in ("batch", "odd", "gray")]
        spool = [(DA((N // 2, cap), np.float32, fill_byte=0), DA((N // 2, cap), np.float32, fill_byte=0))
                 for op in self.ops if op == "stereo"]

        def batch(ptr, stride, n=N):
            nonlocal last_n
            last_n = n
            o = pool[len(outs)]
            outs.append(o)
            ex.extract_batch_device(ptr, n, W, H, stride, H * stride, o[0].ptr, cap, o[1].ptr, o[2].ptr, sync=wait)

        for op, s, nfr in zip(self.ops, self.src, self.frames):
            if op == "batch":
                batch(self.inputs[s].ptr, W, nfr)
            elif op == "odd":
                batch(self.odd.ptr, W + 3, nfr)
            elif op == "gray":                                   # the batch reads what the kernel in front of it writes
                ex.gray_batch_device(N, self.rgb[n_gray & 1].ptr, W, H, 3 * W, 3 * W * H, 3, 0, self.conv.ptr, W, W * H, sync=wait)
                n_gray += 1
                batch(self.conv.ptr, W)
            elif op == "stereo":                                 # reads the last batch's arenas from the main stream
                k, q, c = outs[-1]
                u, z = spool[len(keep)]
                keep.append((u, z))
                ex.stereo_match_batch_device(last_n // 2, k.ptr, cap, q.ptr, c.ptr, 0.5, 300.0, u.ptr, z.ptr, sync=wait)
            elif op == "level":                                  # a read-back (synchronises) of the last batch's pyramid
                res.append(ex.pyramid_level(last_n - 1, NL - 1).copy())
            elif op == "sync":
                ex.synchronize()
        ex.synchronize()
        for o in outs:
            res.extend(a.to_host() for a in o)
            for a in o:
                a.free()
        for u, z in keep:
            res.extend((u.to_host(), z.to_host()))
            u.free()
            z.free()
        return res

    def free(self):
        for a in self.inputs + self.rgb + [self.odd, self.conv]:
            a.free()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_call_sequences_without_waiting(seed):
    ex = oa.ORBextractor(NF, 1.2, NL, 20, 7, max_width=W, max_height=H, max_batch=N, keep_pyramid=True)
    seq = _Seq(ex, seed, 24)
    ref = seq.run(True)
    for ordered in (0, 1):
        ex.set_debug_flag(oa.FLAG_ORDERED, ordered)
        for rep in range(2):
            got = seq.run(False)
            assert len(got) == len(ref)
            for i, (a, b) in enumerate(zip(ref, got)):
                assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (seed, ordered, rep, i, seq.ops)
    counts = [r for r in ref if r.dtype == np.int32 and r.shape == (N,)]
    assert counts and all(c[:8].min() > 0 for c in counts)        # the frames do hold features: the comparison is not empty
    seq.free()
    ex.close()
