#!/usr/bin/env python
"""How much of a training step is the reference's loss glue (ConditionalDDPM.forward around the network call, its
autograd backward, the l2 objective)?  The network is replaced by a one-kernel stand-in so that only the glue is timed:
host time per call (no synchronisation) and GPU time (events) in a free-running loop."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsbdd_amd import synthetic as S   # noqa: E402
from tools.train_step_bench import build, loss_of   # noqa: E402


def main():
    dev = torch.device("cuda:0")
    workload = sys.argv[1] if len(sys.argv) > 1 else "crossdock_fullatom_cond"
    key = "ca" if "ca_" in workload else "fa"
    B = 96 if key == "ca" else 16
    model, cfg, dd = build(workload, dev)
    model.train(True)
    w = torch.nn.Parameter(torch.ones((), device=dev))

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = w

        def forward(self, zl, zp, t, lm, pm):
            return zl * self.w, zp
    real = model.dynamics
    model.dynamics = Fake()
    batches = [(S.load_pocket(key, B, dev), S.anchor_ligand(B, 23, cfg["atom_nf"], dev)) for _ in range(30)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = 0.0
        for pocket, ligand in batches:
            h0 = time.perf_counter()
            loss_of(model(ligand, pocket)).backward()
            th += time.perf_counter() - h0
        t_host = th / len(batches)
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / len(batches)
        print(f"{workload}: glue forward + backward with a one-kernel network: host {t_host * 1e3:.2f} ms per call, "
              f"wall {t_all * 1e3:.2f} ms per call (free-running)")
    fw = 0.0
    for pocket, ligand in batches:
        h0 = time.perf_counter()
        l = loss_of(model(ligand, pocket))
        fw += time.perf_counter() - h0
        l.backward()
    torch.cuda.synchronize()
    print(f"   of which forward (host) {fw / len(batches) * 1e3:.2f} ms")




def sections():
    """Host time of the glue by method (wrapped with timers; nested calls are counted in both)."""
    import collections, functools
    dev = torch.device("cuda:0")
    model, cfg, dd = build("crossdock_fullatom_cond", dev)
    model.train(True)
    w = torch.nn.Parameter(torch.ones((), device=dev))

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = w

        def forward(self, zl, zp, t, lm, pm):
            return zl * self.w, zp
    model.dynamics = Fake()
    acc = collections.OrderedDict()
    names = ["normalize", "delta_log_px", "_draw_t_int", "gamma", "_remove_lig_com", "noised_representation",
             "xh_given_zt_and_epsilon", "sum_except_batch", "SNR", "log_constants_p_x_given_z0", "kl_prior", "_log_ph_given_z0",
             "log_pN", "_to_device", "_hip_device", "inflate_batch_array"]
    for n in names:
        if not hasattr(model, n):
            continue
        f = getattr(model, n)

        def wrap(f, n):
            @functools.wraps(f)
            def g(*a, **k):
                t0 = time.perf_counter()
                r = f(*a, **k)
                acc[n] = acc.get(n, 0.0) + time.perf_counter() - t0
                return r
            return g
        if isinstance(f, torch.nn.Module):
            f.forward = wrap(f.forward, n)
        else:
            object.__setattr__(model, n, wrap(f, n))
    batches = [(S.load_pocket("fa", 16, dev), S.anchor_ligand(16, 23, cfg["atom_nf"], dev)) for _ in range(20)]
    for pocket, ligand in batches[:5]:
        loss_of(model(ligand, pocket)).backward()
    acc.clear()
    t0 = time.perf_counter()
    for pocket, ligand in batches:
        model(ligand, pocket)
    tot = (time.perf_counter() - t0) / len(batches)
    print(f"forward glue host {tot * 1e3:.2f} ms per call; by method (ms):")
    for n, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"   {n}: {v / len(batches) * 1e3:.3f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sections":
        sections()
    else:
        main()
