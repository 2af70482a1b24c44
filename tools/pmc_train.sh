#!/bin/bash
# rocprofv3 --pmc passes over the training-step benchmark (3 steps): counters of the backward edge kernels and the weight-
# gradient GEMM.  Usage: tools/pmc_train.sh TAG  -> gpurun_out/<TAG>_train_pmc_<n>.md
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-trainpmc}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
n=0
for G in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
         "SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  n=$((n+1))
  rm -rf /tmp/pmct_$n
  (cd /tmp && timeout 300 rocprofv3 --pmc $G --kernel-trace -d /tmp/pmct_$n -o p -- python $R/tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 2 --warmup 1 --paths net > /tmp/pmct_$n.log 2>&1)
  DB=$(find /tmp/pmct_$n -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocpd_pmc.py $DB "dsbdd::" > $R/gpurun_out/${TAG}_train_pmc_$n.md; grep "edge_bwd\|wgrad\|edge_wave\|rows_gather\|node_gemm\|loss_cond\|^| kernel\|^|---" $R/gpurun_out/${TAG}_train_pmc_$n.md | cut -c1-300; else echo "no db for group $n"; tail -3 /tmp/pmct_$n.log; fi
  rm -rf /tmp/pmct_$n
done
