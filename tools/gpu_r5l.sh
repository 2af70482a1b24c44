#!/bin/bash
TAG=${1:-r5l}
mkdir -p gpurun_out
python tools/debug_fused_step.py > gpurun_out/${TAG}_debug_fused.log 2>&1; cat gpurun_out/${TAG}_debug_fused.log | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "fused or folded or ragged or bitwise or chain or frame" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log; tail -6 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-emulated-leg > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("value", d["value"], "frac", r["frac"], "whole", r.get("whole_call_frac"), "other", (d.get("other_states") or {}).get("value"))
except Exception as ex:
    print("bench parse failed", ex)
PY
bash tools/prof_short.sh ${TAG}_T50
head -14 gpurun_out/${TAG}_T50_call_sequence.md
