#!/bin/bash
# The whole GPU test suite, then the profile sets of the three measured configurations:
#   tools/all_profiles.sh rNN   ->  gpurun_out/rNN_*, rNN_c5_*, rNN_c2_*  (copy what is to be kept into profiles/)
set -e
P=${1:?usage: all_profiles.sh rNN}
G=$PWD/gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $G/full_tests.log 2>&1 || { tail -40 $G/full_tests.log; exit 1; }
tail -2 $G/full_tests.log
BENCH_ARGS="--config c5" BENCH_FINAL_ARGS="--steps 20 --no-cpu --no-extras" bash tools/round_profiles.sh ${P}_c5
BENCH_ARGS="--config c2" BENCH_FINAL_ARGS="--steps 50 --no-cpu --no-extras" bash tools/round_profiles.sh ${P}_c2
bash tools/round_profiles.sh $P
