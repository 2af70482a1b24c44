#!/bin/bash
# Round 6, session b: split-K edge kernels (csrc/edge_splitk.h) -- micro-benchmark against edge_wave.h (full-atom and C-alpha
# geometry), the parity tests parametrised over the variant, bench legs with every stage on split-K.
TAG=${1:-r6b}
mkdir -p gpurun_out tools/bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_sk.hip -o tools/bin/mbsk 2> gpurun_out/${TAG}_mbsk_build.log || tail -20 gpurun_out/${TAG}_mbsk_build.log
( echo "## full-atom geometry, B = 64"; timeout 300 tools/bin/mbsk 64 20; echo; echo "## C-alpha geometry, B = 32"; timeout 300 tools/bin/mbsk 32 50 ca;
  echo; echo "## full-atom geometry, B = 16"; timeout 300 tools/bin/mbsk 16 30 ) > gpurun_out/${TAG}_mbsk.md 2>&1
cat gpurun_out/${TAG}_mbsk.md
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "sk or granule_variants or batch_composition or forced_multi_tile" > gpurun_out/${TAG}_pytest_sk.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_sk.log
tail -15 gpurun_out/${TAG}_pytest_sk.log
for W in crossdock_ca_cond crossdock_fullatom_cond; do
  for SK in 0 0xFFFFFFFF; do
    DSBDD_SPLITK=$SK timeout 600 python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-emulated-leg --other-steps 3 \
      > gpurun_out/${TAG}_bench_${W}_sk${SK}.json 2> gpurun_out/${TAG}_bench_${W}_sk${SK}.err
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${TAG}_bench_${W}_sk${SK}.json").read().strip().splitlines()[-1])
    o=d.get("other_states") or {}
    print("$W SPLITK=$SK value", round(d["value"],2), "ms/step", round(d["ms_per_step"],1), "dom frac", d["roofline"]["frac"], "| other", o.get("states"), o.get("value"), (o.get("roofline") or {}).get("frac"))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/${TAG}_bench_${W}_sk${SK}.err").read()[-1500:])
PY
  done
done
