#!/bin/bash
# rocprofv3 --pmc passes over a torch-free micro-benchmark binary (seconds per pass).
# Usage: tools/pmc_micro.sh TAG FILTER BINARY "CTR_A CTR_B ..." ["CTR_C ..."] ...   -> gpurun_out/<TAG>_pmc_<n>.md
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; FILTER=$2; BIN=$3; shift 3
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
n=0
for G in "$@"; do
  n=$((n+1))
  rm -rf /tmp/pmcm_$n
  (cd /tmp && timeout 120 rocprofv3 --pmc $G --kernel-trace -d /tmp/pmcm_$n -o p -- $R/$BIN 64 4 > /tmp/pmcm_$n.log 2>&1)
  DB=$(find /tmp/pmcm_$n -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocpd_pmc.py $DB $FILTER > $R/gpurun_out/${TAG}_pmc_$n.md; cat $R/gpurun_out/${TAG}_pmc_$n.md; else echo "no db for group $n"; tail -5 /tmp/pmcm_$n.log; fi
  rm -rf /tmp/pmcm_$n
done
