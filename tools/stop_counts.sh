#!/bin/bash
# Instruction counts of k_fast_tiles when it ends after phase k (diagnostic builds liborbx_stop<k>.so, see
# stop_sweep.sh): one rocprofv3 counter pass per build; differences give the counts per phase.
export TMPDIR=/tmp
for k in 1 2 3 4 5 full; do
  if [ $k = full ]; then unset ORBX_LIB; else export ORBX_LIB=orb_slam2_amd/liborbx_stop$k.so; fi
  OUT=$PWD/gpurun_out/stopc_$k
  mkdir -p $OUT
  timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d $OUT -o pmc -- \
      python3 bench.py --steps 2 --warmup 1 --no-cpu --no-extras > $OUT/bench.log 2>&1 || exit 1
  echo "== stop $k"
  python3 tools/pmc_summary.py $OUT/pmc_counter_collection.csv | grep k_fast_tiles
  rm -rf $OUT
done
