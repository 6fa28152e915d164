#!/bin/bash
# Per-chunk timestamps (ORBX_TRACE=1) of 15 calls of the host batch path: which stage stretches in a slow call.
G=$PWD/gpurun_out
ORBX_TRACE=1 timeout -k 10 300 python tools/host_batch_rate.py 512 15 > $G/r03_jitter_stdout.txt 2> $G/r03_jitter_trace.txt
cat $G/r03_jitter_stdout.txt
grep "host ms" $G/r03_jitter_trace.txt
