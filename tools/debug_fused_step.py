"""Debug: fused keyed step kernel vs the separate launches, piece by piece (GPU)."""
import ctypes as C
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsbdd_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, a, r = 5, 10, 10
nl = torch.tensor([9, 12, 7, 10, 23]); npk = torch.tensor([30, 41, 25, 36, 50])
lm = torch.repeat_interleave(torch.arange(B), nl).to(dev); pm = torch.repeat_interleave(torch.arange(B), npk).to(dev)
dl, dp = 3 + a, 3 + r
z0 = torch.randn(len(lm), dl, device=dev); p0 = torch.randn(len(pm), dp, device=dev); eps = torch.randn(len(lm), dl, device=dev)
xh0 = torch.randn(len(lm), dl, device=dev); com0 = torch.randn(B, 3, device=dev); fixed = (torch.rand(len(lm), device=dev) > 0.5).float()
st = torch.cuda.current_stream(dev).cuda_stream
seed, draw = 1234, 7
def randn(d):
    out = torch.empty(len(lm), dl, device=dev)
    _lib.check(lib.dsbdd_randn_keyed(st, out.data_ptr(), lm.data_ptr(), len(lm), dl, B, 0, None, C.c_uint64(seed), C.c_uint64(d), 0))
    return out
for mode in (0, 1, 2):
    for (al, ce, sg) in ((1.0, 0.0, 1.0), (0.97, 0.13, 0.21)):
        z1, p1 = z0.clone(), p0.clone()
        _lib.check(lib.dsbdd_cond_reverse_update(st, z1.data_ptr(), p1.data_ptr(), eps.data_ptr(), randn(draw).data_ptr(), lm.data_ptr(), pm.data_ptr(),
                                                 len(lm), len(pm), B, a, r, al, ce, sg, 1))
        zu, pu = z1.clone(), p1.clone()
        if mode:
            zk = torch.empty_like(z1)
            n2 = randn(draw + 2)
            _lib.check(lib.dsbdd_cond_repaint_update(st, z1.data_ptr(), p1.data_ptr(), zk.data_ptr(), xh0.data_ptr(), com0.data_ptr(), fixed.data_ptr(),
                                                     randn(draw + 1).data_ptr(), n2.data_ptr() if mode == 2 else None, lm.data_ptr(), pm.data_ptr(), len(lm), len(pm), B, a, r,
                                                     0.8, 0.6, al, 0.33, int(mode == 2), 1))
        z2, p2 = z0.clone(), p0.clone()
        zk2 = torch.empty_like(z2)
        tw = torch.zeros(1, device=dev)
        _lib.check(lib.dsbdd_cond_step_keyed(st, z2.data_ptr(), p2.data_ptr(), eps.data_ptr(), zk2.data_ptr(), xh0.data_ptr(), com0.data_ptr(), fixed.data_ptr(),
                                             lm.data_ptr(), pm.data_ptr(), len(lm), len(pm), B, a, r, al, ce, sg, mode, 0.8, 0.6, 0.33, 1,
                                             C.c_uint64(seed), C.c_uint64(draw), 0, None, tw.data_ptr(), 0.25))
        torch.cuda.synchronize()
        print(f"mode {mode} coef {(al, ce, sg)}: max|dz| {(z1 - z2).abs().max().item():.3e} max|dp| {(p1 - p2).abs().max().item():.3e} "
              f"equal {torch.equal(z1, z2)} {torch.equal(p1, p2)} t_word {tw.item()}")
