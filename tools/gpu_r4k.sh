#!/bin/bash
# free-running leg (the metric's literal sample_given_pocket) under the 16-edge message-stage variants
TAG=${1:-r4k}
mkdir -p gpurun_out
for g in 0 0x3F 0x07 0x38; do
  DSBDD_GRANULE16=$g timeout 300 python bench.py --states free --steps 3 --warmup 1 --no-cpu-baseline --no-other-leg --no-other-workloads > gpurun_out/${TAG}_free_g$g.json 2>> gpurun_out/${TAG}_bench.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_free_g*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f.split("/")[-1], "value %.2f" % d["value"], "ms", round(d["ms_per_step"],1), "frac", round(r["frac"],3), "whole", round(r["whole_call_frac"],3))
PY
tail -3 gpurun_out/${TAG}_bench.err
