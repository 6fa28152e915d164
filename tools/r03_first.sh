#!/bin/bash
# Round 3, first GPU call: the 4K configuration's profile set, then kernel traces with queue ids.
set -e
G=$PWD/gpurun_out
export TMPDIR=/tmp
BENCH_ARGS="--config c5" BENCH_FINAL_ARGS="--steps 20 --no-cpu --no-extras" bash tools/round_profiles.sh r03_c5
mkdir -p $G/q_c3 $G/q_two
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $G/q_c3 -o trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-extras > $G/q_c3/bench.log 2>&1
python3 tools/queue_map.py "$(find $G/q_c3 -name '*kernel_trace.csv' | head -1)" > $G/r03_queue_map_c3.txt
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $G/q_two -o trace -- python3 tools/two_ctx_pattern.py 50 > $G/q_two/run.log 2>&1
python3 tools/queue_map.py "$(find $G/q_two -name '*kernel_trace.csv' | head -1)" > $G/r03_queue_map_two_ctx.txt
cat $G/q_two/run.log | tail -2
rm -rf $G/q_c3 $G/q_two
cat $G/r03_queue_map_c3.txt $G/r03_queue_map_two_ctx.txt
