#!/bin/bash
# first GPU call of round 3: micro-benchmarks (no torch), then the GPU suite and a short bench run
mkdir -p gpurun_out
tools/bin/mfma_shadow > gpurun_out/r3a_mfma_shadow.md 2>&1
for v in base epi epi_noslp; do echo "## $v" >> gpurun_out/r3a_microbench.md; timeout 120 tools/bin/mb_$v 64 20 >> gpurun_out/r3a_microbench.md 2>&1; done
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a_pytest.log
timeout 300 python bench.py --steps 3 --warmup 1 > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
tail -3 gpurun_out/r3a_pytest.log; cat gpurun_out/r3a_mfma_shadow.md; cat gpurun_out/r3a_microbench.md | grep -v "^|--"
