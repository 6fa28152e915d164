#!/bin/bash
# Headline step of several library builds, alternating, three rounds (run on the GPU box through gpurun):
#   gpurun -- 'bash tools/ab_bench.sh orb_slam2_amd/liborbx.so orb_slam2_amd/liborbx_x.so'
for round in 1 2 3; do
  for lib in "$@"; do
    export ORBX_LIB=$lib
    timeout -k 10 200 python bench.py --steps 200 --warmup 5 --no-cpu --no-extras | python -c "
import json,sys
b=json.loads(sys.stdin.read())
print('$lib  %.0f frames/s  %.4f ms/step  blocks median %.4f' % (b['value'], b['ms_per_step'], b['ms_per_step_blocks']['median']))
" || exit 1
  done
done
