#!/usr/bin/env python
"""Host-side profile of the training step on the HIP path (cProfile over 3 steps after 3 warm-up steps): where the Python /
launch time of `ddpm.forward` + `backward` goes.   python tools/train_profile.py [workload] > gpurun_out/<tag>_train_profile.txt"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from diffsbdd_amd import synthetic as S   # noqa: E402
import train_step_bench as tb   # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "crossdock_fullatom_cond"
os.environ["DSBDD_TRAIN"] = "hip"
dev = torch.device("cuda:0")
model, cfg, dd = tb.build(workload, dev)
model.train(True)
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, amsgrad=True, weight_decay=1e-12)
key = "ca" if "ca_" in workload else "fa"
B = 96 if key == "ca" else 16


def step(sync):
    pocket = S.load_pocket(key, B, dev)
    ligand = S.anchor_ligand(B, 23, cfg["atom_nf"], dev)
    opt.zero_grad(set_to_none=True)
    loss = tb.loss_of(model(ligand, pocket))
    if sync:
        torch.cuda.synchronize()
    loss.backward()
    if sync:
        torch.cuda.synchronize()
    opt.step()


for _ in range(3):
    step(True)
torch.cuda.synchronize()
# (i) free-running host time of a step (no synchronisation inside): host-bound if this is the step time
t0 = time.perf_counter()
for _ in range(5):
    step(False)
t_host = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 5
print(f"host time per step without synchronisation {t_host * 1e3:.2f} ms; with the final synchronisation {t_all * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step(False)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:6000])
