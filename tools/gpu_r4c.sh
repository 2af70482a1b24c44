#!/bin/bash
# round 4, session C: 16-edge-granule variant -- parity tests over the variant, bench A/B (forced on / selected stages)
TAG=${1:-r4c}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x -k "rows_spanning or granule or bench_plan or node_chain_kernel" > gpurun_out/${TAG}_pytest16.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/${TAG}_pytest16.log | cut -c1-300
for M in 0 0xFFFFFFFF 0x003F0000; do
  DSBDD_GRANULE16=$M timeout 300 python bench.py --workload crossdock_ca_cond --steps 2 --warmup 1 --no-cpu-baseline --no-other-leg > gpurun_out/${TAG}_ca_g$M.json 2>> gpurun_out/${TAG}_bench.err
done
for M in 0 0x20 0x30 0x003F0020; do
  DSBDD_GRANULE16=$M timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-leg --no-other-workloads > gpurun_out/${TAG}_fa_g$M.json 2>> gpurun_out/${TAG}_bench.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_*_g*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f.split("/")[-1], "value %.2f" % d["value"], "frac", r["frac"] and round(r["frac"], 4), "whole", r.get("whole_call_frac") and round(r["whole_call_frac"], 3))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 gpurun_out/${TAG}_bench.err
