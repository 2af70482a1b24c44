#!/bin/bash
# Round 6, session g: merged partial-vector reduction of the backward edge kernels; wgrad chunk cap A/B; whole GPU suite.
TAG=${1:-r6g}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q > gpurun_out/${TAG}_train_tests.log 2>&1
echo "train tests rc=$?" >> gpurun_out/${TAG}_train_tests.log
tail -4 gpurun_out/${TAG}_train_tests.log
timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 8 --paths net,functions 2>&1 | tail -2 | tee gpurun_out/${TAG}_train_step.md
DSBDD_WGRAD_MAXWG=384 timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 8 --paths net 2>&1 | tail -1
DSBDD_WGRAD_MAXWG=256 timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 8 --paths net 2>&1 | tail -1
timeout 300 python tools/train_step_bench.py --workload crossdock_ca_cond --steps 8 --paths net 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_train_step.md
timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 --deselect tests/test_gpu_train.py > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -16 gpurun_out/${TAG}_pytest.log
