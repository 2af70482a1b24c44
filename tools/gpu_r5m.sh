#!/bin/bash
TAG=${1:-r5m}
mkdir -p gpurun_out
python tools/debug_fused_step.py 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fused or golden or free_running or cond_reverse" 2>&1 | tail -3
for w in 1 2; do
  echo "== DSBDD_TRAIN_WG_PER_CU=$w" >> gpurun_out/${TAG}_train_step.md
  DSBDD_TRAIN_WG_PER_CU=$w timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 8 --paths hip 2>/dev/null >> gpurun_out/${TAG}_train_step.md
  DSBDD_TRAIN_WG_PER_CU=$w timeout 300 python tools/train_step_bench.py --workload crossdock_ca_cond --steps 8 --paths hip 2>/dev/null | tail -1 >> gpurun_out/${TAG}_train_step.md
done
cat gpurun_out/${TAG}_train_step.md
DSBDD_TRAIN_WG_PER_CU=2 timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -3
