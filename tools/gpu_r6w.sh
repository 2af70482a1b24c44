#!/bin/bash
# Round 6, session w: side streams in the training backward (DSBDD_TRAIN_STREAMS bit mask: 1 second coordinate MLP's chain,
# 2 node-level weight gradients, 4 coordinate-stage W2 gradients, 8 message-stage W2 gradient; default 7).
TAG=${1:-r6w}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -x -q > gpurun_out/${TAG}_train_tests.log 2>&1
echo "train tests rc=$?" >> gpurun_out/${TAG}_train_tests.log; tail -3 gpurun_out/${TAG}_train_tests.log
for st in 0 7 0 7; do
  DSBDD_TRAIN_STREAMS=$st timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 10 --paths net 2>/dev/null | tail -1 | sed "s/^/streams=$st /" | tee -a gpurun_out/${TAG}_train_step.md
done
for st in 0 7 0 7; do
  DSBDD_TRAIN_STREAMS=$st timeout 300 python tools/train_step_bench.py --workload crossdock_ca_cond --steps 10 --paths net 2>/dev/null | tail -1 | sed "s/^/streams=$st /" | tee -a gpurun_out/${TAG}_train_step.md
done
bash tools/prof_train.sh ${TAG}
head -1 gpurun_out/${TAG}_train_call_sequence.md
