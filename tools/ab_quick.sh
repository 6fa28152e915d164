#!/bin/bash
# As ab_bench.sh, with the stage times alone on the GPU as well.
for round in 1 2; do
  for lib in "$@"; do
    export ORBX_LIB=$lib
    timeout -k 10 200 python bench.py --steps 200 --warmup 5 --no-cpu --no-extras | python -c "
import json,sys
b=json.loads(sys.stdin.read())
print('$lib  %.4f ms/step  blocks median %.4f' % (b['ms_per_step'], b['ms_per_step_blocks']['median']), {k: round(v,3) for k,v in b['stage_ms_serialised'].items()}, {k: round(v,3) for k,v in b['stage_ms'].items()})
" || exit 1
  done
done
