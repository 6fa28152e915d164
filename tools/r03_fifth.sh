#!/bin/bash
set -e
G=$PWD/gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $G/r03_full_tests.log 2>&1 || { tail -40 $G/r03_full_tests.log; exit 1; }
tail -3 $G/r03_full_tests.log
timeout -k 10 600 python bench.py --no-cpu > $G/r03_bench_a.json 2> $G/r03_bench_a.err
python - <<'PY'
import json
b=json.load(open('gpurun_out/r03_bench_a.json'))
print(b['value'], b['ms_per_step'])
print(json.dumps(b['latency_ms'], indent=1))
print(b['path_valu_issue'])
print(b['host_to_host'])
PY
ORBX_POOL_DEVICES=0 timeout -k 10 300 python bench.py --config c4 --host --steps 3 > $G/r03_host_pool_1.json
ORBX_POOL_DEVICES=0,0 timeout -k 10 300 python bench.py --config c4 --host --steps 3 > $G/r03_host_pool_2.json
cat $G/r03_host_pool_1.json $G/r03_host_pool_2.json | python -c "
import sys,json
for l in sys.stdin:
    b=json.loads(l); print(b['config']['workers'], b['value'], b['ms_per_step'], b['spread_ms'], b['GBs_in'])
"
