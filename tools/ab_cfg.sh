#!/bin/bash
# As ab_bench.sh for another bench configuration: ab_cfg.sh "<bench args>" <steps> lib...
ARGS=$1; STEPS=$2; shift 2
for round in 1 2; do
  for lib in "$@"; do
    export ORBX_LIB=$lib
    timeout -k 10 300 python bench.py $ARGS --steps $STEPS --warmup 3 --no-cpu --no-extras | python -c "
import json,sys
b=json.loads(sys.stdin.read())
print('$lib  %.4f ms/step  blocks median %.4f' % (b['ms_per_step'], b['ms_per_step_blocks']['median']), {k: round(v,3) for k,v in b['stage_ms_serialised'].items()}, {k: round(v,3) for k,v in b['stage_ms'].items()})
" || exit 1
  done
done
