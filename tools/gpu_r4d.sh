#!/bin/bash
TAG=${1:-r4d}
mkdir -p gpurun_out
for v in base xbar; do timeout 120 tools/bin/mb_node_$v 64 30 > gpurun_out/${TAG}_mb_node_$v.md 2>&1; grep "node_chain" gpurun_out/${TAG}_mb_node_$v.md | cut -c1-150; done
timeout 1800 python -m pytest tests -q -m gpu -x > gpurun_out/${TAG}_pytest.log 2>&1; echo "suite rc=$?"; tail -6 gpurun_out/${TAG}_pytest.log | cut -c1-300
for L in 1 0; do DSBDD_LIG_HEAD=$L timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads > gpurun_out/${TAG}_fa_lighead$L.json 2>> gpurun_out/${TAG}_bench.err; done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_fa_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f.split("/")[-1], "value %.2f" % d["value"], "frac", round(r["frac"], 4), "whole", round(r["whole_call_frac"], 3), "other", d["other_states"]["value"])
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 gpurun_out/${TAG}_bench.err
