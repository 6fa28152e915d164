#!/bin/bash
# Time of k_fast_tiles alone on the GPU when it ends after phase k (diagnostic builds liborbx_stop<k>.so):
# 1 staging, 2 + rejection test and queue, 3 + scores, 4 + suppression, 5 + retry pass, full = + ordered output.
for k in 1 2 3 4 5 full; do
  if [ $k = full ]; then unset ORBX_LIB; else export ORBX_LIB=orb_slam2_amd/liborbx_stop$k.so; fi
  timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu --no-extras | python -c "
import json,sys
b=json.loads(sys.stdin.read())
print('stop $k: fast alone %.3f ms   (step %.3f ms)' % (b['stage_ms_serialised']['fast'], b['ms_per_step']))
" || exit 1
done
