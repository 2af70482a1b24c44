#!/bin/bash
# Round 6, session c: call sequences with the split-K kernels (C-alpha x 32 with the per-chain auto rule; the free-running
# full-atom chain with every stage on them), the training step under rocprofv3 (no kernel table since r4m).
TAG=${1:-r6c}
mkdir -p gpurun_out
bash tools/prof_short.sh ${TAG}_ca --workload crossdock_ca_cond --splitk auto
bash tools/prof_short.sh ${TAG}_ca_default --workload crossdock_ca_cond
bash tools/prof_short.sh ${TAG}_free_sk --states free --splitk 0xFFFFFFFF
bash tools/prof_short.sh ${TAG}_free_default --states free
head -3 gpurun_out/${TAG}_ca_call_sequence.md; grep -A12 "^| kernel | grid" gpurun_out/${TAG}_ca_call_sequence.md | head -16
head -1 gpurun_out/${TAG}_free_sk_call_sequence.md; grep -A8 "^| kernel | grid" gpurun_out/${TAG}_free_sk_call_sequence.md | head -12
bash tools/prof_train.sh ${TAG}
head -30 gpurun_out/${TAG}_train_kernel_stats.md; cat gpurun_out/${TAG}_train_under_rocprof.md
for W in crossdock_ca_cond; do
  timeout 600 python bench.py --workload $W --splitk auto --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-emulated-leg --other-steps 3 \
      > gpurun_out/${TAG}_bench_${W}_auto.json 2> gpurun_out/${TAG}_bench_${W}_auto.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${TAG}_bench_${W}_auto.json").read().strip().splitlines()[-1])
    o=d.get("other_states") or {}
    print("$W splitk auto: value", round(d["value"],2), "ms/step", round(d["ms_per_step"],1), "| other", o.get("states"), o.get("value"))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/${TAG}_bench_${W}_auto.err").read()[-1500:])
PY
done
