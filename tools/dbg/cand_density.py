import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import orb_slam2_amd as oa
from orb_slam2_amd import synth
for (w,h,nf,nl) in ((1241,376,2000,8),(3840,2160,8000,12)):
    base = synth.make_batch(1, w, h, 5, 0)
    ex = oa.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=1, keep_pyramid=False)
    ex(base[0])
    tot = 0; px = 0
    for l in range(nl):
        c = ex.debug_read(oa.DBG_CANDIDATES, 0, l)
        n = len(c)//4
        tot += n
    print(w,h,"candidates after NMS per frame", tot, "per Mpx", tot / (w*h*3.2/1e6))
    ex.close()
