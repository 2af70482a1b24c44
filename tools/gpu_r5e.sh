#!/bin/bash
TAG=${1:-r5e}
mkdir -p gpurun_out
for v in sc sc_np sc_nopipeA sc_f406np sc_sgb2np; do
  echo "=== $v" >> gpurun_out/${TAG}_mbe.md
  timeout 120 tools/bin/mbe_$v 64 20 0 2>&1 | grep "emu-6\|whole batch" >> gpurun_out/${TAG}_mbe.md
done
cat gpurun_out/${TAG}_mbe.md
