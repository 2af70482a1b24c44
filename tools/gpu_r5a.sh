#!/bin/bash
# Round 5, first GPU session: emulated-fp32 edge kernels -- micro-benchmark (timing + error vs float64), the gate tests,
# a bench line with the path on, and the whole GPU suite with DSBDD_EMU=6.   Usage: tools/gpu_r5a.sh TAG
TAG=${1:-r5a}
mkdir -p gpurun_out
timeout 300 tools/bin/mb_emu 64 20 > gpurun_out/${TAG}_mb_emu.md 2>&1; echo "mb_emu rc=$?"; cat gpurun_out/${TAG}_mb_emu.md
timeout 900 python -m pytest tests/test_gpu_emu.py -x -q -s --durations=12 > gpurun_out/${TAG}_pytest_emu.log 2>&1; echo "pytest_emu rc=$?"; tail -25 gpurun_out/${TAG}_pytest_emu.log
DSBDD_EMU=6 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads > gpurun_out/${TAG}_bench_emu6.json 2> gpurun_out/${TAG}_bench_emu6.err; echo "bench emu rc=$?"
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_bench_emu6.json").read().strip().splitlines()[-1])
    r = d["roofline"]; print("EMU6 value", d["value"], "avg_launch_ms", r["avg_launch_ms"], "frac(fp32 peak)", r["frac"], "other", d["other_states"] and d["other_states"]["value"])
except Exception as e:
    print("bench parse failed", e)
PY
tail -3 gpurun_out/${TAG}_bench_emu6.err
DSBDD_EMU=6 timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_emu.py > gpurun_out/${TAG}_emu_gate_pytest.log 2>&1; echo "gate rc=$?"; tail -15 gpurun_out/${TAG}_emu_gate_pytest.log
