#!/bin/bash
# Round 6, session l: loss-head A/B tests (incl. SimpleConditionalDDPM).
TAG=${1:-r6l}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py -m gpu -q -k "hip_launches or edge_capacity" > gpurun_out/${TAG}_tests.log 2>&1
echo "rc=$?" >> gpurun_out/${TAG}_tests.log; tail -8 gpurun_out/${TAG}_tests.log
