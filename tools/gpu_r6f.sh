#!/bin/bash
# Round 6, session f: the one-launch-sequence training step -- tests (whole file + the parity test that checks None
# gradients), timing net vs functions, rocprofv3 kernel table of the new path.
TAG=${1:-r6f}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py tests/test_gpu_reference_caller.py -m gpu -x -q -k "train or gradients or loss" > gpurun_out/${TAG}_train_tests.log 2>&1
echo "train tests rc=$?" >> gpurun_out/${TAG}_train_tests.log
tail -12 gpurun_out/${TAG}_train_tests.log
timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 8 --paths net,functions 2>&1 | tail -4 | tee gpurun_out/${TAG}_train_step.md
timeout 300 python tools/train_step_bench.py --workload crossdock_ca_cond --steps 8 --paths net,functions 2>&1 | tail -2 | tee -a gpurun_out/${TAG}_train_step.md
DSBDD_TRAIN_WG_PER_CU=2 timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 8 --paths net 2>&1 | tail -1
bash tools/prof_train.sh ${TAG}
head -45 gpurun_out/${TAG}_train_kernel_stats.md; cat gpurun_out/${TAG}_train_under_rocprof.md
