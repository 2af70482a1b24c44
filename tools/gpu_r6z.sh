#!/bin/bash
# Round 6 closing GPU session: HBM traffic by PMC (bench.py reads profiles/r6_pmc_traffic.json and checks the kernel-source
# hash), the whole GPU suite with durations, the default bench line (all legs: free-running + its split-K variant, emulated,
# C-alpha x 32 / mixed / joint with split-K auto and emulated legs, training step), rocprofv3 kernel stats of the full-length
# command (exact and emulated), call sequences (anchored, free-running, C-alpha with split-K), the training step (timing +
# kernel table), the split-K micro-benchmark, the DSBDD_EMU=6 gate run of the suite.  Usage: tools/gpu_r6z.sh TAG
TAG=${1:-r6z}
mkdir -p gpurun_out tools/bin
bash tools/pmc_traffic.sh r6 > gpurun_out/${TAG}_pmc_traffic.log 2>&1; tail -2 gpurun_out/${TAG}_pmc_traffic.log | cut -c1-300
[ -f gpurun_out/r6_pmc_traffic.json ] && cp gpurun_out/r6_pmc_traffic.json profiles/r6_pmc_traffic.json
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log; tail -4 gpurun_out/${TAG}_pytest.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_bench_default.json").read().strip().splitlines()[-1])
    print("summary:", json.dumps(d["summary"]))
    r = d["roofline"]
    print("value", d["value"], "frac", r["frac"], "whole", r.get("whole_call_frac"), "traffic", r.get("traffic"), (r.get("traffic_source") or "")[:80])
    for w in d.get("other_workloads") or []: print("   ", w["workload"], w.get("pockets"), w.get("value"), w.get("ms_per_step"), (w.get("emulated") or {}).get("value"), w.get("edge_splitk"), w.get("error"))
    print("cpu", (d.get("cpu_baseline") or {}).get("kind"), (d.get("cpu_baseline") or {}).get("value"))
except Exception as ex:
    print("bench parse failed", ex)
PY
tail -4 gpurun_out/${TAG}_bench.err
bash tools/prof_full.sh ${TAG}_T500 --steps 2 --warmup 1
bash tools/prof_full.sh ${TAG}_T500_emu6 --steps 2 --warmup 1 --emulation 6
bash tools/prof_short.sh ${TAG}_T50
bash tools/prof_short.sh ${TAG}_T50_free --states free
bash tools/prof_short.sh ${TAG}_T50_free_sk --states free --splitk 0xFFFFFFFF
bash tools/prof_short.sh ${TAG}_ca --workload crossdock_ca_cond --splitk auto
timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 8 --paths net,functions > gpurun_out/${TAG}_train_step.md 2>/dev/null
timeout 300 python tools/train_step_bench.py --workload crossdock_ca_cond --steps 8 --paths net,functions 2>/dev/null | tail -2 >> gpurun_out/${TAG}_train_step.md
cat gpurun_out/${TAG}_train_step.md
bash tools/prof_train.sh ${TAG}
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_sk.hip -o tools/bin/mbsk 2>/dev/null
( echo "## full-atom geometry, B = 64"; timeout 300 tools/bin/mbsk 64 20; echo; echo "## C-alpha geometry, B = 32"; timeout 300 tools/bin/mbsk 32 50 ca;
  echo; echo "## full-atom geometry, B = 16"; timeout 300 tools/bin/mbsk 16 30 ) > gpurun_out/${TAG}_mbsk.md 2>&1
DSBDD_EMU=6 timeout 2000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_emu.py > gpurun_out/${TAG}_emu_gate_pytest.log 2>&1; echo "gate rc=$?" >> gpurun_out/${TAG}_emu_gate_pytest.log; tail -4 gpurun_out/${TAG}_emu_gate_pytest.log
ls gpurun_out | grep "^${TAG}" | tr '\n' ' '
