#!/bin/bash
# Round 3, second GPU call: the tests touched by the hardening, then the step with 4 and with 8 hardware queues.
set -e
G=$PWD/gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_hardening.py tests/test_zz_no_alloc.py tests/test_gpu_bowmatch.py tests/test_gpu_parity.py tests/test_gpu_local_map.py tests/test_gpu_initmatch.py -m gpu -x -q --timeout 180 --timeout-method thread > $G/r03_second_tests.log 2>&1 || { tail -30 $G/r03_second_tests.log; exit 1; }
tail -3 $G/r03_second_tests.log
for round in 1 2; do
  for q in 4 8; do
    GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --steps 200 --warmup 5 --no-cpu --no-extras | python -c "
import json,sys
b=json.loads(sys.stdin.read())
print('hw queues $q: %.0f frames/s  %.4f ms/step  blocks median %.4f' % (b['value'], b['ms_per_step'], b['ms_per_step_blocks']['median']), {k: round(v,3) for k,v in b['stage_ms'].items()})
"
  done
done
