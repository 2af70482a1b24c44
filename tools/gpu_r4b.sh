#!/bin/bash
# round 4, session B: new tests (RCCL one-rank, driver vs oracle), the default bench line with the new legs,
# launch_floor (guide's XCD barrier), sustained test-set run with rejections, kernel trace of a training step
TAG=${1:-r4b}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_rccl.py tests/test_testset.py -q -m gpu -x > gpurun_out/${TAG}_pytest_new.log 2>&1; echo "new tests rc=$?"; tail -15 gpurun_out/${TAG}_pytest_new.log | cut -c1-300
timeout 60 tools/bin/launch_floor > gpurun_out/${TAG}_launch_floor.md 2>&1; cat gpurun_out/${TAG}_launch_floor.md
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/${TAG}_bench.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    r=d["roofline"]; o=d["other_states"]; c=d["cpu_baseline"]
    print("value", d["value"], "frac", r["frac"], "whole", r["whole_call_frac"])
    print("other", o["value"], o["roofline"]["frac"], o["roofline"]["whole_call_frac"], o["roofline"]["timed_launches"])
    print("cpu", c["kind"], c["value"], c.get("port"), c.get("reference_error"))
    for w in d["other_workloads"] or []: print(w["workload"], w["pockets"], w["value"], w["whole_call_frac"], w["stage_radii"])
except Exception as e:
    print("bench parse failed", e)
PY
timeout 600 python tools/testset_sustained.py 24 500 keyed:0.6 > gpurun_out/${TAG}_testset_sustained.md 2> gpurun_out/${TAG}_testset.err; echo "sustained rc=$?"; head -12 gpurun_out/${TAG}_testset_sustained.md; tail -2 gpurun_out/${TAG}_testset.err
export TMPDIR=/tmp; rm -rf /tmp/tr_prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr_prof -o p -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 3 --paths hip > /tmp/tr_prof.log 2>&1)
DB=$(find /tmp/tr_prof -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB > gpurun_out/${TAG}_train_kernel_stats.md 2>&1; head -40 gpurun_out/${TAG}_train_kernel_stats.md | cut -c1-220; else echo "no db"; tail -5 /tmp/tr_prof.log; fi
