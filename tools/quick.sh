#!/bin/bash
# Parity of the extractor path, then the headline bench without the CPU leg: the loop used while tuning kernels.
set -e
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -3
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu | python -c "
import json,sys
b=json.loads(sys.stdin.read())
print('value %.0f frames/s  %.3f ms/step' % (b['value'], b['ms_per_step']))
print('stage_ms           ', {k: round(v,3) for k,v in b['stage_ms'].items()})
print('stage_ms_serialised', {k: round(v,3) for k,v in b['stage_ms_serialised'].items()})
"
