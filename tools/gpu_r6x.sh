#!/bin/bash
# Round 6, session x: the forward pass keeps z2 of the edge MLPs; the element-wise kernels E / EC replace kernel A.
TAG=${1:-r6x}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_reference_caller.py -m gpu -x -q > gpurun_out/${TAG}_train_tests.log 2>&1
echo "train tests rc=$?" >> gpurun_out/${TAG}_train_tests.log; tail -5 gpurun_out/${TAG}_train_tests.log
for st in 0 1 0 1; do
  DSBDD_TRAIN_STORE_Z2=$st timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 10 --paths net 2>/dev/null | tail -1 | sed "s/^/store_z2=$st /" | tee -a gpurun_out/${TAG}_train_step.md
done
for st in 0 1 0 1; do
  DSBDD_TRAIN_STORE_Z2=$st timeout 300 python tools/train_step_bench.py --workload crossdock_ca_cond --steps 10 --paths net 2>/dev/null | tail -1 | sed "s/^/store_z2=$st /" | tee -a gpurun_out/${TAG}_train_step.md
done
bash tools/prof_train.sh ${TAG}
head -12 gpurun_out/${TAG}_train_kernel_stats.md | cut -c1-150
