#!/bin/bash
TAG=${1:-r4i}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q -m gpu -x -k "node_chain or bench_plan or bitwise or batch_composition or full_size_chain or ragged" > gpurun_out/${TAG}_pytest.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${TAG}_pytest.log | cut -c1-300
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads > gpurun_out/${TAG}_fa.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --workload crossdock_ca_cond --steps 2 --warmup 1 --no-cpu-baseline --no-other-leg --granule16 auto > gpurun_out/${TAG}_ca.json 2>> gpurun_out/${TAG}_bench.err
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_*a.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f.split("/")[-1], "value %.2f" % d["value"], "ms", round(d["ms_per_step"],1), "whole", round(r["whole_call_frac"],3), "other", (d.get("other_states") or {}).get("value"))
PY
