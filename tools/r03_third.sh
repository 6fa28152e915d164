#!/bin/bash
set -e
G=$PWD/gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_cpp_class.py tests/test_gpu_multiproc.py tests/test_gpu_stereo.py tests/test_golden.py -m gpu -x -q --timeout 300 --timeout-method thread > $G/r03_third_tests.log 2>&1 || { tail -40 $G/r03_third_tests.log; exit 1; }
tail -3 $G/r03_third_tests.log
python - <<'PY'
import numpy as np, subprocess, os, sys
sys.path.insert(0, '.')
from orb_slam2_amd import synth
w,h,pairs=1241,376,200
base=synth.make_batch(8,w+32,h,3,900)
fr=np.zeros((2*pairs,h,w),np.uint8)
for p in range(pairs):
    fr[2*p]=base[p%8][:,:w]; fr[2*p+1]=base[p%8][:,7:w+7]
fr.tofile('/tmp/pairs.raw')
for keep in ('1','0'):
    for mode in (1,0,2):
        out=subprocess.run(['tests/cpp/orbextractor_harness','threads','2000','1.2','8',str(w),str(h),str(pairs),'/tmp/pairs.raw','0.5371657','386.1448','/tmp/o.bin',str(mode)],capture_output=True,text=True,env=dict(os.environ,ORBX_KEEP_PYRAMID=keep))
        print('keep_pyramid',keep,out.stdout.strip(),out.stderr[-300:])
PY
python tools/latency_abi.py
