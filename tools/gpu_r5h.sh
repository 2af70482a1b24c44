#!/bin/bash
TAG=${1:-r5h}
mkdir -p gpurun_out
for v in base micro d_all; do for g in 512 256 128; do
  echo "=== $v grid cap $g" >> gpurun_out/${TAG}_mbe.md
  timeout 120 tools/bin/mbe_$v 64 10 0 $g 2>&1 | grep "whole list" >> gpurun_out/${TAG}_mbe.md
done; done
cat gpurun_out/${TAG}_mbe.md
