#!/usr/bin/env python
"""In-kernel timeline of the fused edge kernels (debug build with -DDSBDD_TIMESTAMPS, loaded through
DSBDD_LIB): wall_clock64() marks of the first 64 workgroups of every edge launch of a few eager EGNN calls.
Prints, per launch kind (message stage / coordinate stage), the mean time between marks in microseconds:
entry -> prologue done -> end of each K step -> epilogue done.
Usage: DSBDD_LIB=build_ab/lib_ts.so DSBDD_GRAPH=0 python tools/exp_timestamps.py [--workload crossdock_ca_cond]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="crossdock_ca_cond")
    ap.add_argument("--batch", type=int, default=None)
    a = ap.parse_args()
    arch, key, b0 = bench.WORKLOADS[a.workload]
    B = a.batch or b0
    dev = torch.device("cuda", 0)
    cfg, dd, model = bench.build_model(arch, dev)
    pocket = bench.load_pocket(key, B, dev)
    n_lig = torch.full((B,), 23, dtype=torch.int64)
    model.seed(1)
    model.sample_given_pocket({k: v.clone() for k, v in pocket.items()}, n_lig, timesteps=3)     # warm-up
    eng = model.dynamics.engine()
    cap = 64
    buf = torch.zeros(cap * 1024, dtype=torch.int64, device=dev)
    eng.lib.dsbdd_debug_set_timestamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert eng.lib.dsbdd_debug_set_timestamps(eng.handle, buf.data_ptr(), cap) == 0
    model.sample_given_pocket({k: v.clone() for k, v in pocket.items()}, n_lig, timesteps=4)
    torch.cuda.synchronize()
    ts = buf.cpu().numpy().reshape(cap, 64, 16).astype(np.float64)
    L = cfg["n_layers"]
    for kind, name in ((0, "message stage (GCL)"), (1, "coordinate stage")):
        rows = []
        for li in range(2 * L, cap):                       # skip the first call
            if li % 2 != kind:
                continue
            t = ts[li]
            ok = t[:, 1] > 0                                # workgroups that had a tile
            if not ok.any():
                continue
            d = (t[ok] - t[ok][:, :1]) / 100.0              # 100 MHz -> us since entry
            n_marks = int((t[ok][0] > 0).sum())
            rows.append(np.concatenate([d[:, :n_marks].mean(0), [ok.sum()]]))
        rows = np.array(rows)
        m = rows.mean(0)
        print(f"{name}: {len(rows)} launches, {m[-1]:.0f} of the first 64 workgroups active; "
              f"us since kernel entry at marks [prologue, K steps ..., epilogue]:")
        print("   " + "  ".join(f"{v:6.1f}" for v in m[1:-1]))
    # spread of the entry times over the 64 recorded workgroups (dispatch ramp)
    t0 = ts[2 * L:, :, 0]
    t0 = t0[(t0 > 0).all(1)]
    if len(t0):
        print(f"dispatch ramp: entry of workgroup 63 minus workgroup 0 = {((t0[:, 63] - t0[:, 0]) / 100.0).mean():.2f} us")


if __name__ == "__main__":
    main()
