#!/bin/bash
TAG=${1:-r5d}
mkdir -p gpurun_out
for v in asym2 asym2dyn dyn asym3dyn_nopipeA asym1dyn; do
  echo "=== $v" >> gpurun_out/${TAG}_mbe.md
  timeout 120 tools/bin/mbe_$v 64 20 0 2>&1 | grep "emu-6\|whole batch" >> gpurun_out/${TAG}_mbe.md
done
cat gpurun_out/${TAG}_mbe.md
