#!/bin/bash
# Round 6, session s: the kept-z2 / recompute comparison test and the training tests on the final tree.
TAG=${1:-r6s}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py -m gpu -x -q -s -k "kept_z2 or side_streams" > gpurun_out/${TAG}_z2_tests.log 2>&1
echo "rc=$?" >> gpurun_out/${TAG}_z2_tests.log; grep "worst\|passed\|failed\|rc=" gpurun_out/${TAG}_z2_tests.log | tail -8
