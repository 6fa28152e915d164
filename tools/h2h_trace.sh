#!/bin/bash
# Host batch path inside the benchmark process, per-chunk trace (ORBX_TRACE=1): without and with the copy-engine warm-up.
# The "without" build:  make -C orb_slam2_amd/csrc OUT=../liborbx_nowarm.so EXTRA=-DORBX_DIAG_NO_PIPE_WARM
#   gpurun -- 'bash tools/h2h_trace.sh'   -> gpurun_out/r03_h2h_trace_{before,after}.txt
G=$PWD/gpurun_out
ORBX_LIB=orb_slam2_amd/liborbx_nowarm.so ORBX_TRACE=1 python bench.py --steps 20 --no-cpu > $G/h2h_before.json 2> $G/r03_h2h_trace_before.txt
ORBX_TRACE=1 python bench.py --steps 20 --no-cpu > $G/h2h_after.json 2> $G/r03_h2h_trace_after.txt
for f in before after; do python -c "
import json
b=json.load(open('$G/h2h_$f.json')); print('$f', b['host_to_host']['spread_ms']['calls_ms'])"; done
grep -n "calls [1-9]\.[0-9]*)" $G/r03_h2h_trace_before.txt | head; echo ==; grep -n "calls [1-9]\.[0-9]*)" $G/r03_h2h_trace_after.txt | head
