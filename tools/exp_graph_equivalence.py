#!/usr/bin/env python
"""Check that eager launches, hipGraph replay and a mix of both give the same chain.
Usage: exp_graph_equivalence.py [timesteps] [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(mode, T, B):
    os.environ["DSBDD_GRAPH"] = "0" if mode == "eager" else "1"
    dev = torch.device("cuda", 0)
    cfg, dd, model = bench.build_model("crossdock_fullatom_cond", dev)
    eng = model.dynamics.engine()
    if mode.startswith("mixed"):
        eng.profile(int(mode[5:]), max_launches=8192)
    pocket = bench.load_pocket("fa", B, dev)
    n_lig = torch.full((B,), 23, dtype=torch.int64)
    model.seed(7)
    out, _, mask, _ = model.sample_given_pocket(pocket, n_lig, timesteps=T)
    torch.cuda.synchronize()
    print(mode, "graph stats", eng.graph_stats(), "finite", bool(torch.isfinite(out).all()))
    return out.cpu()


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    ref = run("eager", T, B)
    for mode in ("graph", "mixed3", "mixed8"):
        try:
            out = run(mode, T, B)
            print(f"  {mode}: max |diff| vs eager = {(out - ref).abs().max().item():.3e}")
        except Exception as ex:   # noqa: BLE001
            print(f"  {mode}: FAILED {type(ex).__name__}: {ex}")


if __name__ == "__main__":
    main()
