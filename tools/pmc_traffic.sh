#!/bin/bash
# HBM traffic of the dominant kernel from rocprofv3 PMC counters: FETCH_SIZE and WRITE_SIZE in two
# separate passes (they do not fit one pass; no other trace domains are combined with --pmc), over
# `bench.py --steps 1 --warmup 0 --timesteps 5`.  Writes gpurun_out/<TAG>_pmc_{fetch,write}.md and
# gpurun_out/<TAG>_pmc_traffic.json (units / gfx950 correction as MI355X_MICROARCH.md "HBM" prescribes:
# counter value = KiB; FETCH_SIZE under-counts wide reads 2x on gfx950 -> doubled, an upper bound).
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r3}
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  (cd /tmp && timeout 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o p -- python $R/bench.py --steps 1 --warmup 0 --timesteps 5 --no-cpu-baseline --no-kernel-timing --no-other-leg --no-emulated-leg --no-other-workloads > /tmp/pmc_$C.log 2>&1)
  DB=$(find /tmp/pmc_$C -name "*.db" | head -1)
  if [ -z "$DB" ]; then echo "no db for $C"; tail -3 /tmp/pmc_$C.log; continue; fi
  L=$(echo $C | tr 'A-Z' 'a-z' | sed 's/_size//')
  python $R/tools/rocpd_pmc.py $DB edge_wave --longest 0.9 > $R/gpurun_out/${TAG}_pmc_$L.md
  rm -rf /tmp/pmc_$C
done
python - <<PY
import json, re, sys
sys.path.insert(0, "$R")
from diffsbdd_amd.build import kernel_source_hash
def val(path, col):
    rows = [l for l in open(path) if l.startswith("| \`dsbdd::edge_wave_kernel<256, 0")]
    hdr = [c.strip() for c in open(path).readline().strip().strip("|").split("|")]
    cells = [c.strip() for c in rows[0].strip().strip("|").split("|")]
    return float(cells[hdr.index(col)]), int(cells[1])
f, n = val("$R/gpurun_out/${TAG}_pmc_fetch.md", "FETCH_SIZE")
w, _ = val("$R/gpurun_out/${TAG}_pmc_write.md", "WRITE_SIZE")
out = {"kernel": "edge_wave_kernel<256, MODE_GCL, BPERM>", "launches_averaged": n, "kernel_source_sha16": kernel_source_hash(),
       "FETCH_SIZE_kb": f, "WRITE_SIZE_kb": w, "fetch_bytes_corrected": 2 * f * 1024, "write_bytes": w * 1024,
       "traffic_bytes_per_launch": 2 * f * 1024 + w * 1024,
       "traffic_bytes_per_launch_uncorrected": (f + w) * 1024,
       "dispatch_filter": "duration >= 0.9 x the 90th-percentile duration of the kernel = the launches bench.py times (largest radius)",
       "_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py --steps 1 --warmup 0 "
                   "--timesteps 5; bytes = counter * 1024; FETCH doubled per the gfx950 note of MI355X_MICROARCH.md (upper bound)"}
json.dump(out, open("$R/gpurun_out/${TAG}_pmc_traffic.json", "w"), indent=1)
print(out)
PY
