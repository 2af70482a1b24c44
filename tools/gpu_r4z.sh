#!/bin/bash
# Round 4 closing GPU session: HBM traffic by PMC first (bench.py reads profiles/<tag>_pmc_traffic.json), then the suite,
# the default bench line (all legs), the per-workload lines, call sequences, kernel stats of the full-length command,
# training-step timing, micro-benchmarks.   Usage: tools/gpu_r4z.sh TAG
TAG=${1:-r4z}
mkdir -p gpurun_out
bash tools/pmc_traffic.sh $TAG > gpurun_out/${TAG}_pmc_traffic.log 2>&1; tail -2 gpurun_out/${TAG}_pmc_traffic.log | cut -c1-300
[ -f gpurun_out/${TAG}_pmc_traffic.json ] && cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log; tail -4 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_fullatom_cond_B64_T500.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
for c in 2 0; do
  DSBDD_CONE=$c timeout 300 python bench.py --pockets mixed --steps 2 --warmup 1 --no-cpu-baseline --no-other-leg --no-other-workloads > gpurun_out/${TAG}_bench_mixed_cone$c.json 2>> gpurun_out/${TAG}_bench.err
done
timeout 300 python bench.py --workload crossdock_ca_cond --steps 3 --warmup 1 --no-cpu-baseline --granule16 auto > gpurun_out/${TAG}_bench_ca_cond_B32_T500.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --workload moad_fullatom_joint --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_joint_B64.json 2>> gpurun_out/${TAG}_bench.err
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1], "value %.2f" % d["value"], "frac", r["frac"] and round(r["frac"], 4), "whole", r.get("whole_call_frac") and round(r["whole_call_frac"], 3),
              "traffic", r.get("traffic"), "other", (d.get("other_states") or {}).get("value"), "cpu", (d.get("cpu_baseline") or {}).get("kind"), (d.get("cpu_baseline") or {}).get("value"))
        for w in d.get("other_workloads") or []: print("   ", w["workload"], w["pockets"], round(w["value"], 2))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 gpurun_out/${TAG}_bench.err
bash tools/prof_short.sh ${TAG}_T50
bash tools/prof_short.sh ${TAG}_T50_free --states free
bash tools/prof_full.sh ${TAG}_T500 --steps 2 --warmup 1
bash tools/prof_short.sh ${TAG}_ca --workload crossdock_ca_cond --granule16 auto
timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 5 > gpurun_out/${TAG}_train_step.md 2>/dev/null
timeout 300 python tools/train_step_bench.py --workload crossdock_ca_cond --steps 5 2>/dev/null | tail -2 >> gpurun_out/${TAG}_train_step.md
cat gpurun_out/${TAG}_train_step.md
[ -x tools/bin/mb16 ] && timeout 120 tools/bin/mb16 64 20 > gpurun_out/${TAG}_mb16_fa.md 2>&1
[ -x tools/bin/mb_node_new ] && timeout 120 tools/bin/mb_node_new 64 30 > gpurun_out/${TAG}_microbench_node.md 2>&1
timeout 600 python tools/testset_sustained.py 24 500 keyed:0.6 > gpurun_out/${TAG}_testset_sustained.md 2>> gpurun_out/${TAG}_bench.err
ls gpurun_out | grep "^${TAG}" | tr '\n' ' '
