#!/bin/bash
# Several rocprofv3 counter passes, each in its own run (run on the GPU box through gpurun):
#   gpurun -- 'bash tools/pmc_multi.sh <tag>'
set -e
TAG=${1:-m}
run() { bash tools/pmc_bench.sh ${TAG}_$1 "$2" > gpurun_out/pmc_${TAG}_$1.txt 2>&1; }
run sq  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" &&
run sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" &&
run rd  "FETCH_SIZE TCC_HIT_sum" &&
run wr  "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" &&
run tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" &&
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT"
cat gpurun_out/pmc_${TAG}_*.txt | grep -E "k_blur|k_fast|k_orient|k_resize" 
