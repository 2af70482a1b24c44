#!/bin/bash
# emulated-path variants of the micro-benchmark (compile-time switches of csrc/edge_wave.h)
TAG=${1:-r5b}
mkdir -p gpurun_out
for v in base np nopipeA nopipeA_np f406np nofence_np sgb2np sgb2fnp; do
  echo "=== $v" >> gpurun_out/${TAG}_mbe.md
  timeout 120 tools/bin/mbe_$v 64 20 $([ $v = base ] && echo 1 || echo 0) 2>&1 | grep -v "emu-9\|^$\|exact" >> gpurun_out/${TAG}_mbe.md
done
cat gpurun_out/${TAG}_mbe.md
