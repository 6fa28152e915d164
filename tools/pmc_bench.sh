#!/bin/bash
# One rocprofv3 counter pass over a short benchmark run (run on the GPU box through gpurun):
#   gpurun -- 'bash tools/pmc_bench.sh <tag> "<counters>"'
# Counters go in their own run (no tracing); one pass per call.
set -e
TAG=$1
CTRS=$2
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o pmc -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu --no-extras $BENCH_ARGS > $OUT/bench.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc_counter_collection.csv
