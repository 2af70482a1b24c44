#!/bin/bash
# round 4, session A: the training-kernel tests group by group (a GPU fault in one group must not hide the others),
# the training-step timing, PMC passes over node_chain_kernel (micro-benchmark binary)
TAG=${1:-r4a}
mkdir -p gpurun_out
for k in wgrad hip_linear edge_gcl edge_coord oracle_small full_width bitwise; do
  timeout 600 python -m pytest tests/test_gpu_train.py -q -k $k -s > gpurun_out/${TAG}_train_$k.log 2>&1
  echo "== $k rc=$?"; tail -25 gpurun_out/${TAG}_train_$k.log | cut -c1-400
done
timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 3 > gpurun_out/${TAG}_train_step_fa.md 2>gpurun_out/${TAG}_train_step_fa.err; cat gpurun_out/${TAG}_train_step_fa.md; tail -3 gpurun_out/${TAG}_train_step_fa.err
timeout 300 python tools/train_step_bench.py --workload crossdock_ca_cond --steps 3 > gpurun_out/${TAG}_train_step_ca.md 2>gpurun_out/${TAG}_train_step_ca.err; cat gpurun_out/${TAG}_train_step_ca.md; tail -3 gpurun_out/${TAG}_train_step_ca.err
bash tools/pmc_micro.sh ${TAG}_node node_chain tools/bin/mb_final \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT" \
  "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES GRBM_GUI_ACTIVE" 2>&1 | tail -40
