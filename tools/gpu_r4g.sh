#!/bin/bash
TAG=${1:-r4g}
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu -x > gpurun_out/${TAG}_pytest.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/${TAG}_pytest.log | cut -c1-300
for L in 1 0; do
  DSBDD_FORK=$L timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads > gpurun_out/${TAG}_fa_fork$L.json 2>> gpurun_out/${TAG}_bench.err
  DSBDD_FORK=$L timeout 300 python bench.py --workload crossdock_ca_cond --steps 2 --warmup 1 --no-cpu-baseline --no-other-leg --granule16 auto > gpurun_out/${TAG}_ca_fork$L.json 2>> gpurun_out/${TAG}_bench.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_*_fork*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f.split("/")[-1], "value %.2f" % d["value"], "ms", round(d["ms_per_step"],1), "whole", round(r["whole_call_frac"],3), "other", (d.get("other_states") or {}).get("value"), d["hipgraph"])
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 gpurun_out/${TAG}_bench.err
