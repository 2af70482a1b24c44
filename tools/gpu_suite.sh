#!/bin/bash
# GPU suite + a short benchmark run; outputs under gpurun_out/<TAG>_*.  Usage: tools/gpu_suite.sh TAG [pytest -k expr]
TAG=${1:-suite}; K=${2:-}
mkdir -p gpurun_out
if [ -n "$K" ]; then timeout 1200 python -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/${TAG}_pytest.log 2>&1
else timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; fi
echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
timeout 300 python bench.py --steps 3 --warmup 1 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -15 gpurun_out/${TAG}_pytest.log
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel ms", d["roofline"]["avg_launch_ms"], "free", (d.get("other_states") or {}).get("value"))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/${TAG}_bench.err").read()[-2000:])
PY
