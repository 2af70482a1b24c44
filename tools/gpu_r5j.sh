#!/bin/bash
TAG=${1:-r5j}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_reference_caller.py tests/test_gpu_parity.py tests/test_testset.py tests/test_gpu_fullsize.py -m gpu -q -k "reference or fused or folded or ragged or golden or free_running or bitwise or chain or driver or variants or eight or two_ranks" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log; tail -8 gpurun_out/${TAG}_pytest.log
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
tail -3 gpurun_out/${TAG}_bench.err
bash tools/prof_short.sh ${TAG}_T50
head -5 gpurun_out/${TAG}_T50_call_sequence.md
