"""Randomised parity sweep: the HIP path against the CPU oracle on random image sizes, extractor parameters and
image kinds, through the single-frame entry point (one graph launch) and through the batch entry point (the
batch kernels: streaming resize, frame-major FAST and blur).  Every stage is compared bit-exactly; the first
mismatch is printed with the case's parameters, so that it can be replayed as a test.  Run on the GPU box:
     gpurun -- 'python tools/fuzz_parity.py --cases 300 --seed 1'
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402

import orb_slam2_amd as oa  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402


def random_case(rng, max_w=1600, max_h=1200):
    scale = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.25, 1.3, 1.5, 2.0]))
    nlev = int(rng.integers(1, 11))
    # every level must keep 62 px on each side, and the image must not be much taller than wide
    lo = int(np.ceil(64 * scale ** (nlev - 1))) + 2
    if lo > min(max_w, max_h) - 100:
        nlev = 3
        lo = int(np.ceil(64 * scale ** (nlev - 1))) + 2
    w = int(rng.integers(lo, max(lo + 1, max_w)))
    h = int(rng.integers(lo, max(lo + 1, min(max_h, int(1.4 * w)))))
    nf = int(rng.choice([30, 100, 300, 500, 1000, 1500, 2000, 3000, 6000]))
    ini = int(rng.integers(5, 60))
    mn = int(rng.integers(1, ini + 1))
    kind = str(rng.choice(["textured", "textured", "textured", "noise", "dots", "smooth", "blocks", "flat"]))
    return dict(w=w, h=h, nf=nf, scale=scale, nlev=nlev, ini=ini, mn=mn, kind=kind, seed=int(rng.integers(1 << 30)))


def make_image(c, k=0):
    w, h = c["w"], c["h"]
    rng = np.random.default_rng(c["seed"] + 7919 * k)
    kind = c["kind"]
    if kind == "textured":
        return synth.make_frame(w, h, c["seed"] % 97, k)
    if kind == "noise":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == "dots":
        img = np.full((h, w), int(rng.integers(40, 200)), np.uint8)
        dots = rng.random((h, w)) < rng.choice([0.002, 0.02, 0.1])
        img[dots] = rng.integers(0, 256, int(dots.sum()), dtype=np.uint8)
        return img
    if kind == "smooth":
        y, x = np.mgrid[0:h, 0:w]
        img = 128 + 100 * np.sin(x / rng.uniform(5, 40)) * np.cos(y / rng.uniform(5, 40))
        return np.clip(img + rng.normal(0, rng.uniform(0, 6), (h, w)), 0, 255).astype(np.uint8)
    if kind == "blocks":
        img = np.full((h, w), 90, np.uint8)
        for _ in range(int(w * h / 1500)):
            x0, y0 = int(rng.integers(0, w - 4)), int(rng.integers(0, h - 4))
            img[y0:y0 + int(rng.integers(2, 40)), x0:x0 + int(rng.integers(2, 40))] = int(rng.integers(0, 256))
        return img
    return np.full((h, w), int(rng.integers(0, 256)), np.uint8)   # flat: no keypoint anywhere


def compare(ex, orc, idx, kps, desc, okps, odesc, nlev, lists=True):
    for l in range(nlev):
        if not (ex.pyramid_level(idx, l) == orc.level(l)).all():
            return "pyramid level %d" % l
    for l in range(nlev):
        w, h = orc.level_dims(l)
        if not (ex.debug_read(oa.DBG_BLURRED, idx, l).reshape(h, w) == orc.blurred(l)).all():
            return "blurred level %d" % l
    for l in range(nlev if lists else 0):   # a batch that ran in chunks keeps no per-frame lists
        got = ex.debug_read(oa.DBG_CANDIDATES, idx, l).view(np.int32).reshape(-1, 3)
        exp = orc.candidates(l)
        exp = np.stack([exp["x"], exp["y"], exp["response"]], 1)
        if got.shape != exp.shape or not (got == exp).all():
            return "candidates level %d (%d vs %d)" % (l, len(got), len(exp))
    if len(kps) != len(okps):
        return "keypoint count %d vs %d" % (len(kps), len(okps))
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        if not (kps[f] == okps[f]).all():
            return "keypoint field %s" % f
    if not (desc == odesc).all():
        return "descriptors"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--batch", type=int, default=9, help="frames per batch call (8 or more: frame-major launches; 16 or more: the chunked host path)")
    ap.add_argument("--max-width", type=int, default=1600)
    ap.add_argument("--max-height", type=int, default=1200)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    bad, t0, nkp = 0, time.time(), 0
    for i in range(a.cases):
        c = random_case(rng, a.max_width, a.max_height)
        try:
            ex = oa.ORBextractor(c["nf"], c["scale"], c["nlev"], c["ini"], c["mn"], max_width=c["w"], max_height=c["h"],
                                 max_batch=a.batch)
        except oa.OrbxError as e:
            print("case %d %s: refused at create: %s" % (i, c, e), flush=True)
            continue
        orc = oracle_lib.Oracle(c["nf"], c["scale"], c["nlev"], c["ini"], c["mn"])
        img = make_image(c)
        okps, odesc = orc.extract(img)
        kps, desc = ex(img)
        err = compare(ex, orc, 0, kps, desc, okps, odesc, c["nlev"])
        if err is None and i % 3 == 0:
            # the same frame inside a batch, at the batch's last index
            imgs = [make_image(c, k + 1) for k in range(a.batch - 1)] + [img]
            res = ex.extract_batch(imgs)
            bk, bd = res[-1]
            err = compare(ex, orc, a.batch - 1, bk, bd, okps, odesc, c["nlev"], lists=a.batch < 16)
            if err:
                err = "batch path: " + err
        nkp += len(okps)
        if err:
            bad += 1
            print("MISMATCH case %d %s: %s" % (i, c, err), flush=True)
        elif i % 20 == 0:
            print("case %d ok (%d keypoints so far, %.0f s)" % (i, nkp, time.time() - t0), flush=True)
        ex.close()
    print("fuzz_parity: %d cases, %d mismatches, %d keypoints compared, seed %d" % (a.cases, bad, nkp, a.seed), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
