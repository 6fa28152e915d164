#!/bin/bash
# Kernel trace of a short bench run and the timeline of one steady-state step (tools/step_timeline.py).
# BENCH_ARGS: further bench.py arguments (e.g. "--config c5"); SKIP: FAST launches from the end of the trace to the step shown.
set -e
OUT=$PWD/gpurun_out
D=/tmp/orbx_tl
rm -rf $D && mkdir -p $D
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/bench.py --steps ${STEPS:-60} --warmup 5 --no-cpu --no-extras $BENCH_ARGS > $OUT/timeline_bench.json 2> $OUT/timeline_bench.err
CSV=$(find $D -name '*kernel_trace.csv' | head -1)
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $CSV ${SKIP:-150} > $OUT/step_timeline.txt
cat $OUT/step_timeline.txt
