#!/usr/bin/env python
"""Operator-level profile of one training step on the HIP path (torch.profiler, 2 steps): which ATen ops issue the
small launches.   python tools/train_ops.py [workload] > gpurun_out/<tag>_train_ops.txt"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

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
batches = [(S.load_pocket(key, B, dev), S.anchor_ligand(B, 23, cfg["atom_nf"], dev)) for _ in range(5)]


def step(b):
    pocket, ligand = b
    opt.zero_grad(set_to_none=True)
    tb.loss_of(model(ligand, pocket)).backward()
    opt.step()


for b in batches[:3]:
    step(b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for b in batches[3:]:
        step(b)
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
print("| op | calls per step | self CPU us per step | device us per step |")
print("|---|---|---|---|")
for e in rows[:60]:
    dev_t = getattr(e, "self_device_time_total", None)
    if dev_t is None:
        dev_t = getattr(e, "self_cuda_time_total", 0)
    print(f"| {e.key[:70]} | {e.count / 2:.0f} | {e.self_cpu_time_total / 2:.0f} | {dev_t / 2:.0f} |")
