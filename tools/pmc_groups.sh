#!/bin/bash
# rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only) over a short bench run.
# Usage: tools/pmc_groups.sh TAG FILTER "CTR_A CTR_B ..." ["CTR_C ..."] ...   -> gpurun_out/<TAG>_pmc_<n>.md
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; FILTER=$2; shift 2
export TMPDIR=/tmp
n=0
for G in "$@"; do
  n=$((n+1))
  rm -rf /tmp/pmcg_$n
  (cd /tmp && timeout 200 rocprofv3 --pmc $G --kernel-trace -d /tmp/pmcg_$n -o p -- python $R/bench.py --timesteps 6 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing ${BENCH_ARGS} > /tmp/pmcg_$n.log 2>&1)
  DB=$(find /tmp/pmcg_$n -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocpd_pmc.py $DB $FILTER > $R/gpurun_out/${TAG}_pmc_$n.md; else echo "no db for group $n"; tail -3 /tmp/pmcg_$n.log; fi
  rm -rf /tmp/pmcg_$n
done
