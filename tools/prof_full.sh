#!/bin/bash
# rocprofv3 kernel trace of the benchmark command itself (T = 500 chains, HIP-event timing on, as `python bench.py`
# runs it; without the CPU-baseline and secondary legs) -> per-kernel stats under gpurun_out/.
# Usage: tools/prof_full.sh TAG [extra bench.py args]
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-proffull}; shift
export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
(cd /tmp && timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o p -- python $R/bench.py --no-cpu-baseline --no-other-leg --no-emulated-leg --no-other-workloads "$@" > /tmp/prof_$TAG.log 2>&1)
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
if [ -z "$DB" ]; then echo "no rocpd database"; tail -5 /tmp/prof_$TAG.log; exit 1; fi
grep '^{' /tmp/prof_$TAG.log > $R/gpurun_out/${TAG}_bench_under_rocprof.json
python $R/tools/rocpd_stats.py $DB 30 > $R/gpurun_out/${TAG}_kernel_stats.md
rm -rf /tmp/prof_$TAG
