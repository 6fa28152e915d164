#!/bin/bash
# A/B of library builds on the GPU box: parity of the extractor path, then the headline step, per build.
#   gpurun -- 'bash tools/ab.sh orb_slam2_amd/liborbx.so orb_slam2_amd/liborbx_x.so ...'
for lib in "$@"; do
  echo "== $lib"
  export ORBX_LIB=$lib
  tools/quick.sh || exit 1
done
