#!/bin/bash
# emulated path: diagnostic builds (results are WRONG by construction; timing only) -- what bounds the kernel
TAG=${1:-r5c}
mkdir -p gpurun_out
for v in base d_nobread d_nodma d_noact d_nobar d_nogather d_noepi d_nobread_nodma d_all; do
  echo "=== $v" >> gpurun_out/${TAG}_mbe_diag.md
  timeout 120 tools/bin/mbe_$v 64 20 0 2>&1 | grep "emu-6\|emu-9" | grep "whole\|69\|COORD" >> gpurun_out/${TAG}_mbe_diag.md
done
cat gpurun_out/${TAG}_mbe_diag.md
