#!/bin/bash
# The profile set of a round, on the GPU box through gpurun:  gpurun -- 'bash tools/round_profiles.sh r02'
# Leaves gpurun_out/<tag>_{kernel_stats.csv,traffic.json,valu.json,pmc_sq.txt,bench.json}; copy them into profiles/.
# BENCH_ARGS (environment) is added to every bench.py command (e.g. "--config c5"), BENCH_FINAL_ARGS to the last one only.
# Every profiler pass is its own run of the same short benchmark (`bench.py --steps N --no-cpu --no-extras`).
set -e
TAG=${1:-rXX}
G=$PWD/gpurun_out
bash tools/profile_bench.sh $TAG > $G/${TAG}_profile.log 2>&1
cp "$(find $G/prof_$TAG -name '*kernel_stats*' | head -1)" $G/${TAG}_kernel_stats.csv
rm -rf $G/prof_$TAG
bash tools/traffic_pass.sh $TAG > $G/${TAG}_traffic.log 2>&1
python3 tools/traffic_summary.py $G/traffic_$TAG $G/${TAG}_traffic.json >> $G/${TAG}_traffic.log
rm -rf $G/traffic_$TAG
bash tools/pmc_multi.sh $TAG orbx > $G/${TAG}_pmc_sq.txt 2>&1
python3 tools/valu_summary.py $G/${TAG}_valu.json $G/pmc_${TAG}_sq/pmc_counter_collection.csv $G/pmc_${TAG}_sq2/pmc_counter_collection.csv > $G/${TAG}_valu.log
rm -rf $G/pmc_${TAG}_sq $G/pmc_${TAG}_sq2 $G/pmc_${TAG}_sq3 $G/pmc_${TAG}_sq*.txt
# the bench line reads the counter summaries of the running sources from profiles/
cp $G/${TAG}_traffic.json $G/${TAG}_valu.json profiles/
timeout -k 10 900 python3 bench.py $BENCH_ARGS $BENCH_FINAL_ARGS > $G/${TAG}_bench.json 2> $G/${TAG}_bench.err
echo done
