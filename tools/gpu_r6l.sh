#!/bin/bash
# Round 6, session l: the loss terms around the network call on three HIP launches (csrc/loss_head.h), DSBDD_LOSS=torch vs hip.
TAG=${1:-r6l}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_reference_caller.py tests/test_gpu_parity.py -m gpu -q -k "hip_launches" > gpurun_out/${TAG}_tests.log 2>&1
echo "rc=$?" >> gpurun_out/${TAG}_tests.log; tail -25 gpurun_out/${TAG}_tests.log
