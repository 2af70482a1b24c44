#!/usr/bin/env python
"""Time dsbdd_train_wgrad (split-K TN GEMM + ordered reduction) on the shapes of one training step, under the plan given by
DSBDD_WGRAD_MINKC / DSBDD_WGRAD_MAXWG.  Prints one markdown row per shape: us per call (wgrad + reduction), TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsbdd_amd import _lib   # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
shapes = [(4944, 512, 256), (4944, 256, 512), (4944, 256, 256), (4944, 1024, 256), (4944, 256, 33), (368, 256, 256),
          (24000, 256, 256), (91152, 256, 256), (244690, 256, 256)]
tag = f"minkc {os.environ.get('DSBDD_WGRAD_MINKC', '64')} maxwg {os.environ.get('DSBDD_WGRAD_MAXWG', '768')}"
s = torch.cuda.current_stream().cuda_stream
row = []
for K, M, N in shapes:
    A = torch.randn(K, M, device=dev)
    B = torch.randn(K, N, device=dev)
    Cc = torch.empty(M, N, device=dev)
    nb = lib.dsbdd_train_wgrad_scratch_bytes(K, M, N)
    scr = torch.empty(nb, dtype=torch.uint8, device=dev)
    for _ in range(3):
        lib.dsbdd_train_wgrad(s, A.data_ptr(), M, B.data_ptr(), N, K, M, N, Cc.data_ptr(), scr.data_ptr(), nb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.dsbdd_train_wgrad(s, A.data_ptr(), M, B.data_ptr(), N, K, M, N, Cc.data_ptr(), scr.data_ptr(), nb)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    row.append(f"{us:.1f} ({2.0 * K * M * N / us * 1e-6:.1f})")
if os.environ.get("WGRAD_HEADER"):
    print("| plan | " + " | ".join(f"K={K} {M}x{N}" for K, M, N in shapes) + " |")
    print("|---|" + "---|" * len(shapes))
print(f"| {tag} | " + " | ".join(row) + " |")
