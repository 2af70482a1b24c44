#!/usr/bin/env python
"""Time the reference's training step (lightning_modules.py:337-363: ddpm.forward -> l2 loss -> backward -> AdamW) on the
HIP forward / backward kernels (train_hip.py) against round 3's eager torch path (train_path.py), at the batch sizes the
reference's configs train with (configs/crossdock_fullatom_cond.yml:13: 16; crossdock_ca_cond.yml:13: 96).
Synthetic batch: the 3rfm pocket x B with the anchored 23-atom ligand pose (diffsbdd_amd/synthetic.py), random-init weights.

    python tools/train_step_bench.py [--workload crossdock_fullatom_cond] [--batch 16] [--steps 5]
Prints one markdown table row per path: the step with a synchronisation after every phase (forward / backward / optimiser
times) and the same step in a free-running loop (no synchronisation inside: how a training loop runs it)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsbdd_amd import synthetic as S   # noqa: E402


def build(workload, dev):
    from diffsbdd_amd.conditional_model import ConditionalDDPM
    from diffsbdd_amd.dynamics import EGNNDynamics
    from diffsbdd_amd.en_diffusion import EnVariationalDiffusion
    cfg, dd = S.arch_cfg(workload)
    torch.manual_seed(0)
    dyn = EGNNDynamics(**cfg, device=dev)
    with torch.no_grad():
        for n, p in dyn.named_parameters():
            if n.endswith("coord_mlp.4.weight"):
                p.mul_(300.0)                    # visible coordinate updates with random weights
    cls = ConditionalDDPM if dd["conditional"] else EnVariationalDiffusion
    model = cls(dynamics=dyn, atom_nf=cfg["atom_nf"], residue_nf=cfg["residue_nf"], n_dims=3,
                size_histogram=np.ones((40, 400)), timesteps=dd["timesteps"], noise_schedule=dd["noise_schedule"],
                noise_precision=dd["noise_precision"], loss_type="l2", norm_values=dd["norm_values"]).to(dev)
    return model, cfg, dd


def loss_of(terms):
    # the l2 objective of lightning_modules.py:262-275 up to constants: the network-dependent terms of the 12-tuple
    return sum(torch.as_tensor(terms[i]).float().mean() for i in (1, 2, 4, 5, 6))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="crossdock_fullatom_cond")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--paths", default="net,functions,torch",
                    help="DSBDD_TRAIN values: net (one launch sequence per direction, round 6), functions (= hip: the per-stage autograd "
                         "Functions of rounds 4 - 5), torch (round 3's eager path)")
    ap.add_argument("--fused-adam", action="store_true",
                    help="AdamW(fused=True): torch's single multi-tensor kernel instead of the foreach implementation the reference's "
                         "configure_optimizers call gets by default (a caller-side option, reported separately)")
    ap.add_argument("--bare", action="store_true",
                    help="free-running loop only: dynamics forward + a one-kernel loss + backward + optimiser, inputs prepared "
                         "beforehand -- the step without the reference's loss glue (what the ~340 small torch launches cost)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    key = "ca" if "ca_" in a.workload else "fa"
    B = a.batch or (96 if key == "ca" else 16)
    print(f"| workload | batch | path | ms / training step | forward ms | backward ms | optimiser ms | nodes | edges | ms / step, free-running loop |")
    print(f"|---|---|---|---|---|---|---|---|---|---|")
    for path in a.paths.split(","):
        os.environ["DSBDD_TRAIN"] = path
        model, cfg, dd = build(a.workload, dev)
        model.train(True)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, amsgrad=True, weight_decay=1e-12,
                                **({"fused": True} if a.fused_adam else {}))
        tf = tb = to = 0.0
        n_nodes = n_edges = 0
        for it in range(a.warmup + a.steps):
            pocket = S.load_pocket(key, B, dev)
            ligand = S.anchor_ligand(B, 23, cfg["atom_nf"], dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            loss = loss_of(model(ligand, pocket))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            loss.backward()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            opt.step()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            if it >= a.warmup:
                tf += t1 - t0; tb += t2 - t1; to += t3 - t2
        # the same step as a training loop runs it: no synchronisation between forward, backward and the optimiser (the
        # host issues the next phase while the GPU works), batches prepared beforehand, one synchronisation at the end
        batches = [(S.load_pocket(key, B, dev), S.anchor_ligand(B, 23, cfg["atom_nf"], dev)) for _ in range(a.steps)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for pocket, ligand in batches:
            opt.zero_grad(set_to_none=True)
            loss_of(model(ligand, pocket)).backward()
            opt.step()
        torch.cuda.synchronize()
        t_free = (time.perf_counter() - t0) / a.steps
        if a.bare:
            lm, pm = ligand["mask"], pocket["mask"]
            nl, npk = lm.numel(), pm.numel()
            zs = [(torch.randn(nl, 3 + cfg["atom_nf"], device=dev), torch.randn(npk, 3 + cfg["residue_nf"], device=dev) ,
                   torch.rand(B, 1, device=dev)) for _ in range(a.steps)]
            for z in zs:
                z[1][:, :3] = pocket["x"] / dd["norm_values"][0]; z[0][:, :3] = ligand["x"] / dd["norm_values"][0] + 0.1 * z[0][:, :3]
            for rep in range(2):           # (the first pass warms the allocator up for these shapes)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for zl, zp, tt in zs:
                    opt.zero_grad(set_to_none=True)
                    eps, _ = model.dynamics(zl, zp, tt, lm, pm)
                    (eps ** 2).mean().backward()
                    opt.step()
                torch.cuda.synchronize()
            print(f"bare step ({path}): {(time.perf_counter() - t0) / a.steps * 1e3:.2f} ms", flush=True)
        n_nodes = int(pocket["mask"].numel() + ligand["mask"].numel())
        with torch.no_grad():
            e = model.dynamics.get_edges(ligand["mask"], pocket["mask"], ligand["x"], pocket["x"])
        n_edges = int(e.shape[1])
        k = 1e3 / a.steps
        print(f"| {a.workload} | {B} | {path} | {(tf + tb + to) * k:.2f} | {tf * k:.2f} | {tb * k:.2f} | {to * k:.2f} | "
              f"{n_nodes} | {n_edges} | {t_free * 1e3:.2f} |", flush=True)
        del model, opt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
