#!/bin/bash
# Round 6, session d: after the housekeeping of csrc/edge_wave.h (experiment variants moved to tools/edge_wave_diag.h) --
# HBM traffic by PMC on the new kernel-source hash, the micro-benchmarks (default kernel unchanged?), the whole GPU suite.
TAG=${1:-r6d}
mkdir -p gpurun_out tools/bin
bash tools/pmc_traffic.sh r6 > gpurun_out/${TAG}_pmc_traffic.log 2>&1; tail -2 gpurun_out/${TAG}_pmc_traffic.log | cut -c1-400
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_sk.hip -o tools/bin/mbsk 2>/dev/null
( echo "## full-atom geometry, B = 64"; timeout 300 tools/bin/mbsk 64 20; echo; echo "## C-alpha geometry, B = 32"; timeout 300 tools/bin/mbsk 32 50 ca ) > gpurun_out/${TAG}_mbsk.md 2>&1
cat gpurun_out/${TAG}_mbsk.md
timeout 2000 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -20 gpurun_out/${TAG}_pytest.log
