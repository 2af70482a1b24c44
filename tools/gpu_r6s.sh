#!/bin/bash
# Round 6, session s: pass split in the training forward's coordinate stage.
TAG=${1:-r6s}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_reference_caller.py -m gpu -x -q > gpurun_out/${TAG}_train_tests.log 2>&1
echo "rc=$?" >> gpurun_out/${TAG}_train_tests.log; tail -3 gpurun_out/${TAG}_train_tests.log
for i in 1 2 3; do
timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 10 --paths net 2>/dev/null | tail -1 | tee -a gpurun_out/${TAG}_train_step.md
done
bash tools/prof_train.sh ${TAG}
grep "edge_wave_kernel\|total kernel" gpurun_out/${TAG}_train_kernel_stats.md | cut -c1-150
