#!/bin/bash
# Kernel-trace profile of the default benchmark workload; run on the GPU box through gpurun:
#   gpurun -- 'bash tools/profile_bench.sh <tag>'
# Leaves the per-kernel summary in gpurun_out/prof_<tag>/ (copy what should be kept into profiles/).
set -e
TAG=${1:-run}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- \
    python3 bench.py --steps 5 --warmup 2 --no-cpu --no-extras $BENCH_ARGS > $OUT/bench.log 2>&1
tail -2 $OUT/bench.log
find $OUT -name '*kernel_stats*' | head
