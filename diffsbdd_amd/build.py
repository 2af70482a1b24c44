"""In-tree build of libdiffsbdd_hip.so (hipcc, gfx950 only) and of the oracle's
native pieces.  `python -m diffsbdd_amd.build` or `__graft_entry__.build()`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "engine.hip")
OUT = os.path.join(HERE, "libdiffsbdd_hip.so")
def deps():
    """Every source the library is built from (all of csrc/ + the public header)."""
    csrc = os.path.join(HERE, "csrc")
    return sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))) + \
        [os.path.join(ROOT, "include", "diffsbdd_hip.h")]


def kernel_source_hash(names=("edge_wave.h", "edge_mlp.h", "common.h")):
    """sha256 / 16 hex digits of the sources of the dominant kernel (edge_wave_kernel): what a PMC measurement of that
    kernel is valid for (tools/pmc_traffic.sh stamps it into profiles/<tag>_pmc_traffic.json, bench.py compares)."""
    import hashlib
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(HERE, "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.isfile(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def up_to_date():
    if not os.path.isfile(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(d) <= t for d in deps())


def build(force=False, verbose=True, extra_flags=()):
    if not force and up_to_date():
        if verbose:
            print(f"[build] {OUT} is up to date")
        return OUT
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wall", "-Wno-unused-function", *extra_flags, SRC, "-o", OUT]
    if verbose:
        print("[build]", " ".join(cmd))
    subprocess.run(cmd, check=True, cwd=ROOT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
