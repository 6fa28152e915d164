#!/bin/bash
# Kernel trace of a short serialised bench run and the pyramid launches of one batch (tools/pyramid_launches.py).
set -e
OUT=$PWD/gpurun_out
D=/tmp/orbx_pl
rm -rf $D && mkdir -p $D
cd /tmp && export TMPDIR=/tmp
export ORBX_NO_OVERLAP=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-extras $BENCH_ARGS > $OUT/pl_bench.json 2> $OUT/pl_bench.err
CSV=$(find $D -name '*kernel_trace.csv' | head -1)
python3 $GRAFT_REPO_ROOT/tools/pyramid_launches.py $CSV > $OUT/pyramid_launches.txt
cat $OUT/pyramid_launches.txt
