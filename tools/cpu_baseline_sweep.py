#!/usr/bin/env python
"""Thread-count sweep of bench.py's cpu_baseline leg (the oracle = CPU port of the reference path) on
this host: one table row per torch thread count.  With --reference (only where /root/reference exists,
i.e. the build container) the reference's own ConditionalDDPM.sample_p_zs_given_zt is timed next to it
on the same inputs, to show that the port and the reference cost the same.
Usage: cpu_baseline_sweep.py [--batch 16] [--threads 1,8,16,32,64] [--reference]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from diffsbdd_amd import synthetic  # noqa: E402


def reference_step_time(arch, key, b, n_lig, steps):
    from oracle import ref_shim
    dyn_mod, en_mod, cond_mod, _ = ref_shim.import_reference()
    cfg, dd = synthetic.arch_cfg(arch)
    dyn = dyn_mod.EGNNDynamics(**{k: v for k, v in cfg.items()}, device="cpu")
    dyn.load_state_dict(synthetic.random_state_dict(cfg, 0))
    m = cond_mod.ConditionalDDPM(dynamics=dyn, atom_nf=cfg["atom_nf"], residue_nf=cfg["residue_nf"], n_dims=3,
                                 size_histogram=np.ones((4, 8)), timesteps=dd["timesteps"],
                                 noise_schedule=dd["noise_schedule"], noise_precision=dd["noise_precision"],
                                 loss_type="l2", norm_values=dd["norm_values"])
    pocket = bench.load_pocket(key, b, "cpu")
    _, pocket = m.normalize(pocket=pocket)
    xh_p = torch.cat([pocket["x"], pocket["one_hot"]], 1)
    lm = torch.repeat_interleave(torch.arange(b), n_lig)
    z = torch.randn(b * n_lig, 3 + cfg["atom_nf"])
    z[:, :3], xh_p[:, :3] = m.remove_mean_batch(z[:, :3], xh_p[:, :3], lm, pocket["mask"])   # ligand-COM-free state
    T, ts = dd["timesteps"], []
    with torch.no_grad():
        for i, s in enumerate(range(T - 1, T - 2 - steps, -1)):
            sa, ta = torch.full((b, 1), float(s)) / T, torch.full((b, 1), float(s + 1)) / T
            t0 = time.perf_counter()
            z, xh_p = m.sample_p_zs_given_zt(sa, ta, z, xh_p, lm, pocket["mask"])
            if i:
                ts.append(time.perf_counter() - t0)
    return float(np.mean(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--threads", default="1,8,16,32,64")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--reference", action="store_true")
    a = ap.parse_args()
    arch, key = "crossdock_fullatom_cond", "fa"
    print(f"host cores: {os.cpu_count()}; {arch}, batch {a.batch}, {a.steps} timed reverse steps after 1 warm-up")
    print("| torch threads | port s/step | port ligands/s (x501 calls) |" + (" reference s/step |" if a.reference else ""))
    print("|---|---|---|" + ("---|" if a.reference else ""))
    for th in [int(x) for x in a.threads.split(",")]:
        if th > (os.cpu_count() or 1):
            continue
        r = bench.cpu_baseline(arch, key, a.batch, 23, 501, steps=a.steps, max_threads=th)
        t_step = a.batch / (r["value"] * 501)
        row = f"| {th} | {t_step:.3f} | {r['value']:.5f} |"
        if a.reference:
            torch.set_num_threads(th)
            row += f" {reference_step_time(arch, key, a.batch, 23, a.steps):.3f} |"
        print(row, flush=True)


if __name__ == "__main__":
    main()
