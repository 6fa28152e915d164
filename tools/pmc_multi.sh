#!/bin/bash
# Several rocprofv3 counter passes, each in its own run (run on the GPU box through gpurun):
#   gpurun -- 'bash tools/pmc_multi.sh <tag> [kernel-filter]'
set -e
TAG=${1:-m}
FILT=${2:-orbx}
run() { bash tools/pmc_bench.sh ${TAG}_$1 "$2" > gpurun_out/pmc_${TAG}_$1.txt 2>&1; }
run sq  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" &&
run sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" &&
run sq3 "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_INSTS_VALU"
cat gpurun_out/pmc_${TAG}_*.txt | grep -E "$FILT"
