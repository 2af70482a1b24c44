#!/bin/bash
# Round 6, session a: the new parity tests (schedule ends, heterogeneous batches, joint RePaint jump), the whole GPU suite with
# durations, and the default bench line (new: summary key, cpu baseline at batch 64 x 5 steps, 3-chain secondary legs).
TAG=${1:-r6a}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_schedule_ends.py tests/test_chem.py -m gpu -x -q -s > gpurun_out/${TAG}_new_tests.log 2>&1
echo "new tests rc=$?" >> gpurun_out/${TAG}_new_tests.log
tail -5 gpurun_out/${TAG}_new_tests.log
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 --deselect tests/test_gpu_schedule_ends.py > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -40 gpurun_out/${TAG}_pytest.log
( time timeout 900 python bench.py --steps 5 --warmup 2 ) > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 1500 gpurun_out/${TAG}_bench.json; tail -5 gpurun_out/${TAG}_bench.err
