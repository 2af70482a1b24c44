#!/bin/bash
# One GPU session: the GPU suite, the headline bench, the heterogeneous-pocket lines (cone auto / forced / off),
# the C-alpha and joint configs.  Usage: tools/gpu_round.sh TAG [skip-tests]
TAG=${1:-round}; SKIP=${2:-}
mkdir -p gpurun_out
if [ -z "$SKIP" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
  tail -6 gpurun_out/${TAG}_pytest.log
fi
timeout 400 python bench.py --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_fullatom_cond_B64_T500.json 2> gpurun_out/${TAG}_bench.err
for c in 1 2 0; do
  DSBDD_CONE=$c timeout 300 python bench.py --pockets mixed --steps 2 --warmup 1 --no-cpu-baseline --no-other-leg > gpurun_out/${TAG}_bench_mixed_cone$c.json 2>> gpurun_out/${TAG}_bench.err
done
timeout 300 python bench.py --workload crossdock_ca_cond --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_ca_cond_B32_T500.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --workload moad_fullatom_joint --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_joint_B64.json 2>> gpurun_out/${TAG}_bench.err
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1], "value %.2f" % d["value"], "frac", r["frac"] and round(r["frac"], 4), "whole", r.get("whole_call_frac") and round(r["whole_call_frac"], 3),
              "share", r.get("kernel_share_of_wall") and round(r["kernel_share_of_wall"], 3), "plan", r.get("stage_radii"), "other", (d.get("other_states") or {}).get("value"),
              "cpu", (d.get("cpu_baseline") or {}).get("value"), "config0", ((d.get("cpu_baseline") or {}).get("config0") or {}).get("value"))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -5 gpurun_out/${TAG}_bench.err
