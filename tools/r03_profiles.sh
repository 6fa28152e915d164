#!/bin/bash
# Round 3: the whole GPU test suite, then the profile sets of the three measured configurations.
set -e
G=$PWD/gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $G/r03_full_tests.log 2>&1 || { tail -40 $G/r03_full_tests.log; exit 1; }
tail -2 $G/r03_full_tests.log
BENCH_ARGS="--config c5" BENCH_FINAL_ARGS="--steps 20 --no-cpu --no-extras" bash tools/round_profiles.sh r03_c5
BENCH_ARGS="--config c2" BENCH_FINAL_ARGS="--steps 50 --no-cpu --no-extras" bash tools/round_profiles.sh r03_c2
bash tools/round_profiles.sh r03
