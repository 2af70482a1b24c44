#!/usr/bin/env python
"""What the ROCm GEMM libraries (through torch.matmul / addmm, fp32) reach on the node-GEMM shapes, next to
dsbdd_node_linear.  us per launch over 200 back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsbdd_amd import _lib

lib = _lib.load()
d = torch.device("cuda:0")
s = torch.cuda.current_stream(d).cuda_stream
torch.backends.cuda.matmul.allow_tf32 = False


def timeit(fn):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(200):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 200


print("| M | K | N | ours us | torch.addmm us | ours TFLOP/s | library TFLOP/s |")
print("|---|---|---|---|---|---|---|")
for M in (3926, 11545, 19776, 65536):
    for K, N in ((512, 256), (256, 256), (256, 512), (256, 1024)):
        A = torch.randn(M, K, device=d)
        W = torch.randn(K, N, device=d) * 0.05
        b = torch.zeros(N, device=d)
        C = torch.empty(M, N, device=d)
        ours = timeit(lambda: _lib.check(lib.dsbdd_node_linear(s, A.data_ptr(), K, K, None, 0, 0, W.data_ptr(), N,
                                                                b.data_ptr(), None, 0, C.data_ptr(), N, M, N, 0)))
        libt = timeit(lambda: torch.addmm(b, A, W, out=C))
        fl = 2.0 * M * K * N
        print(f"| {M} | {K} | {N} | {ours:.1f} | {libt:.1f} | {fl / ours / 1e6:.1f} | {fl / libt / 1e6:.1f} |")
