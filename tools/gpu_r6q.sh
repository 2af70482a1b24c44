#!/bin/bash
# Round 6, session q: split-K cap of the weight-gradient GEMM on the final training path (DSBDD_WGRAD_MAXWG).
TAG=${1:-r6q}
mkdir -p gpurun_out
for mw in 256 384 512 256 384 512; do
  DSBDD_WGRAD_MAXWG=$mw timeout 300 python tools/train_step_bench.py --workload crossdock_fullatom_cond --steps 10 --paths net 2>/dev/null | tail -1 | sed "s/^/maxwg=$mw /" | tee -a gpurun_out/${TAG}_wgrad_cap.md
done
for mw in 256 384 512; do
  DSBDD_WGRAD_MAXWG=$mw timeout 300 python tools/train_step_bench.py --workload crossdock_ca_cond --steps 10 --paths net 2>/dev/null | tail -1 | sed "s/^/maxwg=$mw /" | tee -a gpurun_out/${TAG}_wgrad_cap.md
done
