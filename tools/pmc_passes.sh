#!/bin/bash
# Several rocprofv3 --pmc passes (one counter group each) over a short bench run;
# writes gpurun_out/<tag>_pmc_<group>.md via tools/rocpd_pmc.py.  Usage: pmc_passes.sh TAG [filter]
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-pmc}; FILTER=${2:-edge_wave}
export TMPDIR=/tmp
declare -A G
G[vmem]="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES"
G[lds]="SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS"
# (TA_* / TCP_* groups hung rocprofv3 on this pool for > 10 min: not collected)
for g in "${!G[@]}"; do
  rm -rf /tmp/pmc_$g
  (cd /tmp && timeout 150 rocprofv3 --pmc ${G[$g]} --kernel-trace -d /tmp/pmc_$g -o p -- python $R/bench.py --timesteps 6 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > /tmp/pmc_$g.log 2>&1)
  DB=$(find /tmp/pmc_$g -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocpd_pmc.py $DB $FILTER > $R/gpurun_out/${TAG}_pmc_$g.md; else echo "no db for $g"; tail -3 /tmp/pmc_$g.log; fi
  rm -rf /tmp/pmc_$g
done
