#!/bin/bash
# rocprofv3 kernel trace of a short benchmark run (T = 50 chains) -> per-kernel stats and the dispatch
# sequence of one EGNN call under gpurun_out/.  Usage: tools/prof_short.sh TAG [extra bench.py args]
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-prof}; shift
export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o p -- python $R/bench.py --timesteps 50 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-leg --no-emulated-leg "$@" > /tmp/prof_$TAG.log 2>&1)
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
if [ -z "$DB" ]; then echo "no rocpd database"; tail -5 /tmp/prof_$TAG.log; exit 1; fi
python $R/tools/rocpd_stats.py $DB 30 > $R/gpurun_out/${TAG}_kernel_stats.md
python $R/tools/rocpd_sequence.py $DB 3 > $R/gpurun_out/${TAG}_call_sequence.md
rm -rf /tmp/prof_$TAG
