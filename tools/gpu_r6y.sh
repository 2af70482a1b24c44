#!/bin/bash
# Round 6, session y (after the closing session): the gradient-assembly tables by value (training tests + timing), the
# assertions that are conditional under the DSBDD_EMU gate run, the training leg of bench.py.
TAG=${1:-r6y}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py tests/test_gpu_reference_caller.py -m gpu -x -q -k "train or gradients or loss" > gpurun_out/${TAG}_train_tests.log 2>&1
echo "train tests rc=$?" >> gpurun_out/${TAG}_train_tests.log; tail -3 gpurun_out/${TAG}_train_tests.log
DSBDD_EMU=6 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_schedule_ends.py -m gpu -q -k "granule_variants or splitk_auto" > gpurun_out/${TAG}_emu_gate_fix.log 2>&1
echo "gate-fix rc=$?" >> gpurun_out/${TAG}_emu_gate_fix.log; tail -3 gpurun_out/${TAG}_emu_gate_fix.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_schedule_ends.py tests/test_chem.py -m gpu -q -k "granule_variants or splitk_auto or virtual" > gpurun_out/${TAG}_exact.log 2>&1
echo "exact rc=$?" >> gpurun_out/${TAG}_exact.log; tail -3 gpurun_out/${TAG}_exact.log
timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 8 --paths net,functions 2>/dev/null | tail -2 | tee gpurun_out/${TAG}_train_step.md
bash tools/prof_train.sh ${TAG}
head -12 gpurun_out/${TAG}_train_kernel_stats.md | cut -c1-160
