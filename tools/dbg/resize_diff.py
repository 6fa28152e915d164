import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import orb_slam2_amd as oa
from orb_slam2_amd import synth
import oracle_lib as oracle
w, h = 640, 480
img = synth.make_frame(w, h, 1, 0)
ex = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h)
ex(img)
orc = oracle.Oracle(1000, 1.2, 8, 20, 7)
orc.extract(img)
for l in range(1, 8):
    got = ex.pyramid_level(0, l); exp = orc.level(l)
    bad = np.argwhere(got != exp)
    print("level", l, got.shape, "mismatches", len(bad))
    if len(bad):
        print(" rows", np.unique(bad[:, 0])[:20], " cols", np.unique(bad[:, 1])[:20])
        for (r, c) in bad[:8]:
            print("  ", r, c, got[r, c], exp[r, c])
        break
