#!/bin/bash
# One rocprofv3 counter pass over one of the rate tools (run on the GPU box through gpurun):
#   gpurun -- 'bash tools/pmc_tool.sh <tag> tools/projection_rate.py "<counters>" [kernel-filter]'
set -e
TAG=$1
TOOL=$2
CTRS=$3
FILT=${4:-orbx}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o pmc -- python3 $TOOL > $OUT/tool.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc_counter_collection.csv | grep -E "$FILT"
rm -rf $OUT
