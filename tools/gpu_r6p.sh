#!/bin/bash
# Round 6, session p: hardware counters of the split-K edge kernels beside the default ones on the micro-benchmark's launches
# (instruction mix, matrix-pipe busy, waits, L2 hit / miss) -- the "what bounds it" evidence of DESIGN 5 "split-K".
TAG=${1:-r6p}
mkdir -p gpurun_out tools/bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_sk.hip -o tools/bin/mbsk 2>/dev/null
bash tools/pmc_micro.sh ${TAG}_sk edge_ tools/bin/mbsk \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" > gpurun_out/${TAG}_pmc_all.log 2>&1
tail -40 gpurun_out/${TAG}_pmc_all.log | cut -c1-400
