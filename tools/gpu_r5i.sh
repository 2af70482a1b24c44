#!/bin/bash
TAG=${1:-r5i}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=20 > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log; tail -40 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    r = d["roofline"]; e = d.get("emulated") or {}
    print("value", d["value"], "frac", r["frac"], "whole", r.get("whole_call_frac"), "traffic_src", r.get("traffic_source"))
    print("workload:", d["config"]["workload"][:120])
    print("other", (d.get("other_states") or {}).get("value"), "emulated", e.get("value"), (e.get("roofline") or {}).get("frac"), (e.get("roofline") or {}).get("algorithmic_fp32_vs_exact_peak"))
    for w in d.get("other_workloads") or []: print("   ", w["workload"], w["pockets"], round(w["value"], 2))
    print("cpu", (d.get("cpu_baseline") or {}).get("kind"), (d.get("cpu_baseline") or {}).get("value"))
except Exception as ex:
    print("bench parse failed", ex)
PY
tail -3 gpurun_out/${TAG}_bench.err
