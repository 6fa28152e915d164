"""Deterministic synthetic textured frames for parity tests and the benchmark.

8-bit gray.  A frame is: 128 + four octaves of bilinearly upsampled uniform noise
(cells of 64/16/4/2 px, amplitudes 64/48/32/20) + W*H/4096 filled random rectangles (side 8..64,
gray uniform 0..255) drawn in order, clamped to 0..255; then one horizontal band (rows
0.45H..0.55H) is overwritten with 128 +- 3 noise, so that some FAST cells find nothing at
the initial threshold and some find nothing at the minimum threshold either (this exercises
the per-cell threshold fallback of ComputeKeyPointsOctTree, reference
src/ORBextractor.cc:903-907).

All randomness is a counter-based splitmix64 hash of (seed, stream, index), so a frame
depends only on (config_id, frame_idx, w, h) and is identical on every machine.
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix(z):
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _u01(seed, stream, n):
    """n uniform doubles in [0,1) for the given (seed, stream)."""
    with np.errstate(over="ignore"):
        base = _mix(np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD6E8FEB86659FD93)))
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = _mix(base + idx * _GOLD)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _octave(seed, stream, w, h, cell, amp):
    gw, gh = w // cell + 2, h // cell + 2
    g = (_u01(seed, stream, gw * gh).reshape(gh, gw) * 2.0 - 1.0) * amp
    ys = (np.arange(h) + 0.5) / cell
    xs = (np.arange(w) + 0.5) / cell
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    top = g[y0][:, x0] * (1 - fx) + g[y0][:, x0 + 1] * fx
    bot = g[y0 + 1][:, x0] * (1 - fx) + g[y0 + 1][:, x0 + 1] * fx
    return top * (1 - fy) + bot * fy


def frame_seed(config_id, frame_idx):
    return (0x9E3779B97F4A7C15 ^ ((int(config_id) << 32) | int(frame_idx))) & 0xFFFFFFFFFFFFFFFF


def make_frame(w, h, config_id=0, frame_idx=0, flat_band=True):
    """Return one (h, w) uint8 synthetic frame."""
    seed = frame_seed(config_id, frame_idx)
    img = np.full((h, w), 128.0)
    img += _octave(seed, 1, w, h, 64, 64.0)
    img += _octave(seed, 2, w, h, 16, 48.0)
    img += _octave(seed, 3, w, h, 4, 32.0)
    img += _octave(seed, 6, w, h, 2, 20.0)
    img = np.clip(np.rint(img), 0, 255)
    nrect = max(1, (w * h) // 4096)
    r = _u01(seed, 4, nrect * 5).reshape(nrect, 5)
    for k in range(nrect):
        rw = 8 + int(r[k, 0] * 57)
        rh = 8 + int(r[k, 1] * 57)
        x = int(r[k, 2] * max(1, w - rw))
        y = int(r[k, 3] * max(1, h - rh))
        img[y:y + rh, x:x + rw] = float(int(r[k, 4] * 256))
    if flat_band:
        y0, y1 = int(0.45 * h), int(0.55 * h)
        if y1 > y0:
            nb = (y1 - y0) * w
            band = 125 + np.floor(_u01(seed, 5, nb) * 7).reshape(y1 - y0, w)
            # the left quarter of the band is exactly flat: no corner at any threshold
            band[:, : w // 4] = 128
            img[y0:y1] = band
    return img.astype(np.uint8)


def make_batch(n, w, h, config_id=0, first_idx=0):
    """Return (n, h, w) uint8; consecutive frame indices (a stereo pair = two of them)."""
    out = np.empty((n, h, w), dtype=np.uint8)
    for i in range(n):
        out[i] = make_frame(w, h, config_id, first_idx + i)
    return out
