#!/bin/bash
# A/B of alternative builds of libdiffsbdd_hip.so on the GPU box:
#   tools/ab_libs.sh build_ab/libA.so build_ab/libB.so ...
# prints ms/chain (T=50) and the node-GEMM dispatch durations of one EGNN block for each.
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
for lib in default "$@"; do
  if [ "$lib" != default ]; then export DSBDD_LIB=$R/$lib; else unset DSBDD_LIB; fi
  echo "== $lib"
  python $R/bench.py --timesteps 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/chain', round(d['ms_per_step'],1), 'gcl ms', round(d['roofline']['avg_launch_ms'],4))"
  rm -rf /tmp/prof_ab; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_ab -o ab -- python $R/bench.py --timesteps 10 --no-cpu-baseline --no-kernel-timing > /tmp/ab.log 2>&1)
  DB=$(find /tmp/prof_ab -name "*.db" | head -1)
  python $R/tools/rocpd_sequence.py $DB 3 | grep -A40 "^| kernel | grid" | grep "node_gemm\|edge_wave\|agg_complete" | head -8
done
