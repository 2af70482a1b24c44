#!/bin/bash
# Round 6, session u: kernel B with the row-gathered P/Q loads one accumulator row ahead.
TAG=${1:-r6u}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q > gpurun_out/${TAG}_train_tests.log 2>&1
echo "train tests rc=$?" >> gpurun_out/${TAG}_train_tests.log; tail -3 gpurun_out/${TAG}_train_tests.log
timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 10 --paths net 2>/dev/null | tail -1 | tee gpurun_out/${TAG}_train_step.md
timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 10 --paths net 2>/dev/null | tail -1 | tee -a gpurun_out/${TAG}_train_step.md
bash tools/prof_train.sh ${TAG}
grep "edge_bwd_a\|edge_bwd_b\|total kernel" gpurun_out/${TAG}_train_kernel_stats.md | cut -c1-150
