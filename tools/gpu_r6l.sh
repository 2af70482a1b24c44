#!/bin/bash
# Round 6, session l: fused loss head + one-launch edge capacity; training tests, parity tests that use edge_capacity, timing.
TAG=${1:-r6l}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_train.py tests/test_gpu_reference_caller.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/${TAG}_tests.log 2>&1
echo "rc=$?" >> gpurun_out/${TAG}_tests.log; tail -5 gpurun_out/${TAG}_tests.log
for i in 1 2 3; do
timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 10 --paths net --bare 2>/dev/null | tail -2 | tee -a gpurun_out/${TAG}_train_step.md
done
timeout 300 python tools/train_step_bench.py --workload crossdock_ca_cond --steps 10 --paths net --bare 2>/dev/null | tail -2 | tee -a gpurun_out/${TAG}_train_step.md
