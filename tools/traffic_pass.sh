#!/bin/bash
# HBM traffic of every kernel of the benchmark step: two rocprofv3 counter passes (FETCH_SIZE and
# WRITE_SIZE cannot share a pass), each in its own run; run on the GPU box through gpurun:
#   gpurun -- 'bash tools/traffic_pass.sh <tag>'
# then, here:  python3 tools/traffic_summary.py gpurun_out/traffic_<tag> profiles/<name>.json
set -e
TAG=${1:-run}
OUT=$PWD/gpurun_out/traffic_$TAG
mkdir -p $OUT/rd $OUT/wr
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/rd -o pmc -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu --no-extras $BENCH_ARGS > $OUT/rd/bench.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/wr -o pmc -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu --no-extras $BENCH_ARGS > $OUT/wr/bench.log 2>&1
ls $OUT/rd $OUT/wr
