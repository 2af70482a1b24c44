#!/bin/bash
# The round's closing GPU session: suite + benches (tools/gpu_round.sh), call sequence and kernel stats of a short run,
# kernel stats of the full-length benchmark command under rocprofv3, HBM traffic by PMC, the micro-benchmarks.
# Usage: tools/gpu_final.sh TAG
TAG=${1:-final}
mkdir -p gpurun_out
bash tools/gpu_round.sh $TAG
bash tools/prof_short.sh ${TAG}_T50
bash tools/prof_full.sh ${TAG}_T500 --steps 2 --warmup 1
bash tools/pmc_traffic.sh $TAG
bash tools/prof_short.sh ${TAG}_ca --workload crossdock_ca_cond
[ -x tools/bin/mfma_shadow ] && timeout 60 tools/bin/mfma_shadow > gpurun_out/${TAG}_mfma_shadow.md 2>&1
[ -x tools/bin/mb_final ] && timeout 120 tools/bin/mb_final 64 40 > gpurun_out/${TAG}_microbench.md 2>&1
ls gpurun_out | grep "^${TAG}" | tr '\n' ' '
