#!/bin/bash
# Round 6, session j: the side-stream join in front of the message stage's W2 gradient (per-set events for the scratch).
TAG=${1:-r6j}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_reference_caller.py -m gpu -q -x > gpurun_out/${TAG}_tests.log 2>&1
echo "rc=$?" >> gpurun_out/${TAG}_tests.log; tail -3 gpurun_out/${TAG}_tests.log
for i in 1 2 3; do
timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 10 --paths net 2>/dev/null | tail -1 | tee -a gpurun_out/${TAG}_train_step.md
done
timeout 300 python tools/train_step_bench.py --workload crossdock_ca_cond --steps 10 --paths net 2>/dev/null | tail -1 | tee -a gpurun_out/${TAG}_train_step.md
bash tools/prof_train.sh ${TAG}
