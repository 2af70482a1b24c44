#!/bin/bash
TAG=${1:-r5g}
mkdir -p gpurun_out
for v in micro micro_np noslp_base; do
  echo "=== $v" >> gpurun_out/${TAG}_mbe.md
  timeout 120 tools/bin/mbe_$v 64 20 $([ $v = micro ] && echo 1 || echo 0) 2>&1 | grep "emu-6\|emu-9\|whole batch\|emulated, \|exact" >> gpurun_out/${TAG}_mbe.md
done
cat gpurun_out/${TAG}_mbe.md
