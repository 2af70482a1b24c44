#!/usr/bin/env python
"""Launch time of the node GEMM (dsbdd_node_linear) against the number of rows: where does a launch that fills
only part of the chip spend its time?  Prints avg us per launch (torch.cuda events over 200 back-to-back launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsbdd_amd import _lib

lib = _lib.load()
d = torch.device("cuda:0")
s = torch.cuda.current_stream(d).cuda_stream
print("| K | N | M | workgroups | us/launch | TFLOP/s |")
print("|---|---|---|---|---|---|")
for K, N in ((256, 256), (512, 256), (256, 512), (256, 1024)):
    W = torch.randn(K, N, device=d) * 0.05
    b = torch.zeros(N, device=d)
    for M in (128, 512, 2048, 4096, 8192, 12288, 16384, 19776, 32768, 65536):
        A = torch.randn(M, K, device=d)
        C = torch.empty(M, N, device=d)
        run = lambda: _lib.check(lib.dsbdd_node_linear(s, A.data_ptr(), K, K, None, 0, 0, W.data_ptr(), N, b.data_ptr(),
                                                       None, 0, C.data_ptr(), N, M, N, 1))
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(200):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        ct = 4 if N >= 512 else 2
        wgs = ((M + 127) // 128) * (N // (32 * ct))
        print(f"| {K} | {N} | {M} | {wgs} | {us:.1f} | {2.0 * M * K * N / us / 1e6:.1f} |")
