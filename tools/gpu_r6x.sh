#!/bin/bash
# Round 6, session x: stored z2 + element-wise kernels E / EC; the side-stream join moved in front of the node gathers.
TAG=${1:-r6x}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_reference_caller.py -m gpu -x -q > gpurun_out/${TAG}_train_tests.log 2>&1
echo "train tests rc=$?" >> gpurun_out/${TAG}_train_tests.log; tail -5 gpurun_out/${TAG}_train_tests.log
for st in 0 7 0 7; do
  DSBDD_TRAIN_STREAMS=$st timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 10 --paths net 2>/dev/null | tail -1 | sed "s/^/streams=$st /" | tee -a gpurun_out/${TAG}_train_step.md
done
for st in 0 7 0 7; do
  DSBDD_TRAIN_STREAMS=$st timeout 300 python tools/train_step_bench.py --workload crossdock_ca_cond --steps 10 --paths net 2>/dev/null | tail -1 | sed "s/^/streams=$st /" | tee -a gpurun_out/${TAG}_train_step.md
done
bash tools/prof_train.sh ${TAG}
head -1 gpurun_out/${TAG}_train_call_sequence.md
