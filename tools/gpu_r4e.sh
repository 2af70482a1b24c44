#!/bin/bash
TAG=${1:-r4e}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -q -x > gpurun_out/${TAG}_pytest_train.log 2>&1; echo "train tests rc=$?"; tail -5 gpurun_out/${TAG}_pytest_train.log | cut -c1-300
timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 5 --paths hip > gpurun_out/${TAG}_train_step_fa.md 2>/dev/null; cat gpurun_out/${TAG}_train_step_fa.md
timeout 300 python tools/train_step_bench.py --workload crossdock_ca_cond --steps 5 --paths hip > gpurun_out/${TAG}_train_step_ca.md 2>/dev/null; tail -1 gpurun_out/${TAG}_train_step_ca.md
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x -k "golden or bench_plan or sample_given or free_running or full_atom_chains" > gpurun_out/${TAG}_pytest_head.log 2>&1; echo "head tests rc=$?"; tail -4 gpurun_out/${TAG}_pytest_head.log | cut -c1-300
for L in 1 0; do DSBDD_LIG_HEAD=$L timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads > gpurun_out/${TAG}_fa_lighead$L.json 2>> gpurun_out/${TAG}_bench.err; done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_fa_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f.split("/")[-1], "value %.2f" % d["value"], "ms", round(d["ms_per_step"],1), "other", d["other_states"]["value"])
    except Exception as e:
        print(f, "FAILED", e)
PY
export TMPDIR=/tmp; rm -rf /tmp/tr_prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr_prof -o p -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 3 --paths hip > /tmp/tr_prof.log 2>&1)
DB=$(find /tmp/tr_prof -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB > gpurun_out/${TAG}_train_kernel_stats.md 2>&1; head -16 gpurun_out/${TAG}_train_kernel_stats.md | cut -c1-200; fi
