#!/usr/bin/env python
"""Experiment: does running the sampling batch as S independent sub-batches on S HIP
streams (one engine each, one host thread each) beat one batch on one stream?
The sub-batches' kernels can fill each other's tail rounds / under-filled grids.
Usage: exp_two_streams.py [--streams 2] [--batch 64] [--timesteps 50]"""
import argparse
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--timesteps", type=int, default=50)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    S, B = a.streams, a.batch // a.streams
    models, pockets, streams = [], [], []
    for i in range(S):
        cfg, dd, m = bench.build_model("crossdock_fullatom_cond", dev)
        models.append(m)
        pockets.append(bench.load_pocket("fa", B, dev))
        streams.append(torch.cuda.Stream(dev))
    n_lig = torch.full((B,), 23, dtype=torch.int64)

    def run(i, seed):
        with torch.cuda.stream(streams[i]):
            models[i].seed(seed, sample_offset=i * B)
            pk = {k: v.clone() for k, v in pockets[i].items()}
            models[i].sample_given_pocket(pk, n_lig, timesteps=a.timesteps)

    def all_once(seed):
        th = [threading.Thread(target=run, args=(i, seed)) for i in range(S)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize(dev)

    all_once(1)
    all_once(2)
    t0 = time.perf_counter()
    for r in range(a.reps):
        all_once(10 + r)
    dt = (time.perf_counter() - t0) / a.reps
    print(f"streams={S} sub-batch={B}: {dt * 1e3:.1f} ms per {a.batch}-ligand chain (T={a.timesteps}) "
          f"= {a.batch / dt:.1f} ligands/s")


if __name__ == "__main__":
    main()
