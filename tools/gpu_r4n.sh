#!/bin/bash
# forward-cone rule: cone on (DSBDD_CONE=2) vs off (0) for batches of 64 with g distinct pockets
TAG=${1:-r4n}
mkdir -p gpurun_out
run() {  # name, extra args
  for c in 0 2; do
    DSBDD_CONE=$c timeout 300 python bench.py $2 --steps 2 --warmup 1 --no-cpu-baseline --no-other-leg --no-other-workloads --no-kernel-timing > gpurun_out/${TAG}_$1_cone$c.json 2>> gpurun_out/${TAG}_bench.err
  done
}
run g01 "--pockets same"
run g08 "--pockets grouped --n-same 57"
run g12 "--pockets grouped --n-same 53"
run g16 "--pockets grouped --n-same 49"
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_g*_cone*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "value %.2f" % d["value"], "ms", round(d["ms_per_step"],1))
PY
