#!/bin/bash
# Round 5 closing GPU session: HBM traffic by PMC first (bench.py reads profiles/<tag>_pmc_traffic.json and checks the kernel
# source hash), the whole GPU suite, the default bench line (all legs), the gate run of the suite with DSBDD_EMU=6, kernel
# stats of the full-length command (exact and emulated), call sequences, training step, micro-benchmarks.  Usage: tools/gpu_r5z.sh TAG
TAG=${1:-r5z}
mkdir -p gpurun_out
bash tools/pmc_traffic.sh $TAG > gpurun_out/${TAG}_pmc_traffic.log 2>&1; tail -2 gpurun_out/${TAG}_pmc_traffic.log | cut -c1-300
[ -f gpurun_out/${TAG}_pmc_traffic.json ] && cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log; tail -4 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench_fullatom_cond_B64_T500.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_bench_fullatom_cond_B64_T500.json").read().strip().splitlines()[-1])
    r = d["roofline"]; e = d.get("emulated") or {}
    print("value", d["value"], "frac", r["frac"], "whole", r.get("whole_call_frac"), "traffic", r.get("traffic"), r.get("traffic_source"))
    print("workload:", d["config"]["workload"][:120])
    print("other", (d.get("other_states") or {}).get("value"), "emulated", e.get("value"), (e.get("roofline") or {}).get("frac"), (e.get("roofline") or {}).get("algorithmic_fp32_vs_exact_peak"))
    for w in d.get("other_workloads") or []: print("   ", w["workload"], w.get("pockets"), w.get("value"), w.get("ms_per_step"), (w.get("roofline") or {}).get("frac"), w.get("error"))
    print("cpu", (d.get("cpu_baseline") or {}).get("kind"), (d.get("cpu_baseline") or {}).get("value"))
except Exception as ex:
    print("bench parse failed", ex)
PY
tail -3 gpurun_out/${TAG}_bench.err
bash tools/prof_full.sh ${TAG}_T500 --steps 2 --warmup 1
bash tools/prof_full.sh ${TAG}_T500_emu6 --steps 2 --warmup 1 --emulation 6
bash tools/prof_short.sh ${TAG}_T50
bash tools/prof_short.sh ${TAG}_T50_free --states free
timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 8 --paths hip > gpurun_out/${TAG}_train_step.md 2>/dev/null
timeout 300 python tools/train_step_bench.py --workload crossdock_ca_cond --steps 8 --paths hip 2>/dev/null | tail -1 >> gpurun_out/${TAG}_train_step.md
cat gpurun_out/${TAG}_train_step.md
[ -x tools/bin/mb_emu ] && timeout 120 tools/bin/mb_emu 64 20 > gpurun_out/${TAG}_mb_emu.md 2>&1
[ -x tools/bin/mb_emu_ph ] && { timeout 120 tools/bin/mb_emu_ph 64 5 0 | grep -A9 "phase clocks"; timeout 120 tools/bin/mb_emu_ph 64 5 0 256 | grep -A9 "phase clocks"; } > gpurun_out/${TAG}_emu_phase_clocks.md 2>&1
[ -x tools/bin/emu_phase_probe ] && timeout 120 tools/bin/emu_phase_probe > gpurun_out/${TAG}_emu_phase_probe.md 2>&1
# matrix-pipe / instruction counters of the exact and the emulated edge kernels side by side (micro-benchmark launches)
[ -x tools/bin/mb_emu ] && bash tools/pmc_micro.sh ${TAG}_emu edge_wave tools/bin/mb_emu "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" > /dev/null 2>&1
DSBDD_EMU=6 timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_emu.py > gpurun_out/${TAG}_emu_gate_pytest.log 2>&1; echo "gate rc=$?" >> gpurun_out/${TAG}_emu_gate_pytest.log; tail -4 gpurun_out/${TAG}_emu_gate_pytest.log
ls gpurun_out | grep "^${TAG}" | tr '\n' ' '
