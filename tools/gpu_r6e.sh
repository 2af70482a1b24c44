#!/bin/bash
# Round 6, session e: the training step as one launch sequence (csrc/train_net.h, train_net.py) -- gradient tests, timing
# against the per-stage Functions (DSBDD_TRAIN=functions), then the rest of the GPU suite.
TAG=${1:-r6e}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q -k "dynamics_training or bitwise or input_gradients" -s > gpurun_out/${TAG}_train_tests.log 2>&1
echo "train tests rc=$?" >> gpurun_out/${TAG}_train_tests.log
tail -30 gpurun_out/${TAG}_train_tests.log
for MODE in net functions; do
  DSBDD_TRAIN=$MODE timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 8 --paths hip 2>&1 | tail -2
done
DSBDD_TRAIN=functions DSBDD_TRAIN_WG_PER_CU=2 timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 8 --paths hip 2>&1 | tail -1
DSBDD_TRAIN_WG_PER_CU=2 timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 8 --paths hip 2>&1 | tail -1
timeout 300 python tools/train_step_bench.py --workload crossdock_ca_cond --steps 8 --paths hip 2>&1 | tail -1
timeout 2000 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_train.py > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -8 gpurun_out/${TAG}_pytest.log
