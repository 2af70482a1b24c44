#!/bin/bash
# rocprofv3 kernel trace of the training-step benchmark (HIP path, 3 timed + 2 warm-up steps) -> per-kernel stats.
# Usage: tools/prof_train.sh TAG [workload]
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-train}; W=${2:-crossdock_fullatom_cond}
export TMPDIR=/tmp
rm -rf /tmp/tr_$TAG
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/tr_$TAG -o p -- python $R/tools/train_step_bench.py --workload $W --steps 3 --paths net > /tmp/tr_$TAG.log 2>&1)
DB=$(find /tmp/tr_$TAG -name "*.db" | head -1)
if [ -z "$DB" ]; then echo "no rocpd database"; tail -5 /tmp/tr_$TAG.log; exit 1; fi
grep "^| cross" /tmp/tr_$TAG.log > $R/gpurun_out/${TAG}_train_under_rocprof.md
python $R/tools/rocpd_stats.py $DB 40 > $R/gpurun_out/${TAG}_train_kernel_stats.md
python $R/tools/rocpd_sequence.py $DB 2 tn_pack_kernel > $R/gpurun_out/${TAG}_train_call_sequence.md
rm -rf /tmp/tr_$TAG
