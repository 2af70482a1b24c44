"""GPU tests of the training-step kernels (csrc/train.h through the C-ABI `dsbdd_train_*`).

Layer 1: every autograd Function of diffsbdd_amd/train_hip.py against the same operation written with torch tensor
operations in float64 (forward values and every input / parameter gradient).
Layer 2: the whole denoiser under autograd (`EGNNDynamics.forward` in training mode) against the ORACLE differentiated by
autograd on the CPU -- the literal reference graph -- on the small architectures and on crossdock_fullatom_cond at
B = 8 (E > 40 k, H = 256), every parameter gradient at 1e-4 of that gradient's largest entry.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import egnn_oracle as eo
from oracle import weights as W

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def make_dynamics(cfg, sd):
    from diffsbdd_amd.dynamics import EGNNDynamics
    m = EGNNDynamics(**cfg, device=dev())
    m.load_state_dict(sd)
    return m


def problem(cfg, n_lig, n_poc, seed, spread=3.0):
    g = torch.Generator().manual_seed(seed)
    B = len(n_lig)
    ml = torch.repeat_interleave(torch.arange(B), torch.tensor(n_lig))
    mp = torch.repeat_interleave(torch.arange(B), torch.tensor(n_poc))
    a, r = cfg["atom_nf"], cfg["residue_nf"]
    xl = torch.cat([torch.randn(len(ml), 3, generator=g) * spread * 0.5, torch.randn(len(ml), a, generator=g)], 1)
    xp = torch.cat([torch.randn(len(mp), 3, generator=g) * spread, torch.randn(len(mp), r, generator=g)], 1)
    t = torch.rand(B, 1, generator=g)
    return xl, xp, t, ml, mp


@pytest.mark.parametrize("K,M,N", [(1000, 256, 256), (37, 20, 10), (5000, 256, 512), (513, 129, 256), (2, 4, 4),
                                     (70000, 256, 256)])
def test_wgrad_and_colsum_vs_torch(K, M, N):
    from diffsbdd_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(K + M + N)
    A = torch.randn(K, M, generator=g).to(dev())
    B = torch.randn(K, N, generator=g).to(dev())
    C_ = torch.empty(M, N, device=dev())
    nb = lib.dsbdd_train_wgrad_scratch_bytes(K, M, N)
    scr = torch.empty(nb, dtype=torch.uint8, device=dev())
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.dsbdd_train_wgrad(s, A.data_ptr(), M, B.data_ptr(), N, K, M, N, C_.data_ptr(), scr.data_ptr(), nb))
    ref = A.double().t() @ B.double()
    assert rel_err(C_, ref) < 2e-6, rel_err(C_, ref)
    C2 = torch.empty_like(C_)
    _lib.check(lib.dsbdd_train_wgrad(s, A.data_ptr(), M, B.data_ptr(), N, K, M, N, C2.data_ptr(), scr.data_ptr(), nb))
    assert torch.equal(C_, C2)                       # ordered reduction: bitwise reproducible
    out = torch.empty(M, device=dev())
    nb2 = 4 * ((K + 31) // 32) * M
    scr2 = torch.empty(nb2, dtype=torch.uint8, device=dev())
    _lib.check(lib.dsbdd_train_colsum(s, A.data_ptr(), M, K, M, out.data_ptr(), scr2.data_ptr(), nb2))
    assert rel_err(out, A.double().sum(0)) < 2e-6


@pytest.mark.parametrize("H,E,e_upd", [(256, 160000, 30000), (192, 30857, 17228), (256, 91152, 11000), (128, 50000, 49000)])
def test_wgrad_on_an_edge_prefix_fits_the_scratch_of_the_full_list(H, E, e_upd):
    """the coordinate stage's backward runs the weight-gradient GEMM on the ligand-row PREFIX
    (K = e_upd < E) with a scratch sized for E.  The split-K plan is not monotonic in K (K = 30 000 needs 188 chunks,
    K = 160 000 only 186 at 256 x 256), so the bound must cover every prefix: `dsbdd_train_wgrad_scratch_bytes(E)` does,
    a call whose plan would not fit returns DSBDD_ERR_CAPACITY instead of writing past the end, and guard words behind
    the scratch stay untouched."""
    from diffsbdd_amd import _lib
    lib = _lib.load()
    nb = lib.dsbdd_train_wgrad_scratch_bytes(E, H, H)
    assert lib.dsbdd_train_wgrad_plan_bytes(e_upd, H, H) <= nb
    g = torch.Generator().manual_seed(E + e_upd)
    A = torch.randn(e_upd, H, generator=g).to(dev())
    B = torch.randn(e_upd, H, generator=g).to(dev())
    guard = 4096
    scr = torch.full((nb + guard,), 0xA5, dtype=torch.uint8, device=dev())
    C_ = torch.empty(H, H, device=dev())
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.dsbdd_train_wgrad(s, A.data_ptr(), H, B.data_ptr(), H, e_upd, H, H, C_.data_ptr(), scr.data_ptr(), nb))
    torch.cuda.synchronize()
    assert bool((scr[nb:] == 0xA5).all())                      # nothing written behind the bound
    assert rel_err(C_, A.double().t() @ B.double()) < 2e-6
    small = lib.dsbdd_train_wgrad_plan_bytes(e_upd, H, H) - 4
    assert lib.dsbdd_train_wgrad(s, A.data_ptr(), H, B.data_ptr(), H, e_upd, H, H, C_.data_ptr(), scr.data_ptr(), small) == _lib.ERR_CAPACITY


@pytest.mark.parametrize("M,K,N,bias", [(777, 256, 256, True), (300, 10, 20, True), (1500, 512, 256, True),
                                         (64, 129, 256, True), (900, 256, 129, False), (500, 256, 1024, False)])
def test_hip_linear_forward_backward(M, K, N, bias):
    from diffsbdd_amd.train_hip import HipLinear
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(dev()).requires_grad_(True)
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev()).requires_grad_(True)
    b = torch.randn(N, generator=g).to(dev()).requires_grad_(True) if bias else None
    gy = torch.randn(M, N, generator=g).to(dev())
    y = HipLinear.apply(x, Wt, b)
    y.backward(gy)
    xd, Wd = x.detach().double().requires_grad_(True), Wt.detach().double().requires_grad_(True)
    bd = b.detach().double().requires_grad_(True) if bias else None
    yr = F.linear(xd, Wd, bd)
    yr.backward(gy.double())
    assert rel_err(y, yr) < 3e-6
    assert rel_err(x.grad, xd.grad) < 3e-6
    assert rel_err(Wt.grad, Wd.grad) < 3e-6
    if bias:
        assert rel_err(b.grad, bd.grad) < 3e-6


@pytest.mark.parametrize("M,H", [(777, 256), (300, 192), (37, 128)])
def test_node_mlp_function(M, H):
    """NodeMLP (GCL.node_model, egnn_new.py:53-58, as one autograd node) vs the same expression in fp64 torch."""
    from diffsbdd_amd.train_hip import NodeMLP
    g = torch.Generator().manual_seed(M + H)
    mk = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(dev()).requires_grad_(True)
    h, agg = mk(M, H), mk(M, H)
    W1, b1, W2, b2 = mk(H, 2 * H, sc=(2 * H) ** -0.5), mk(H), mk(H, H, sc=H ** -0.5), mk(H)
    gy = torch.randn(M, H, generator=g).to(dev())
    y = NodeMLP.apply(h, agg, W1, b1, W2, b2)
    y.backward(gy)
    leaves = [t.detach().double().requires_grad_(True) for t in (h, agg, W1, b1, W2, b2)]
    hd, ad, W1d, b1d, W2d, b2d = leaves
    yr = hd + F.linear(F.silu(F.linear(torch.cat((hd, ad), 1), W1d, b1d)), W2d, b2d)
    yr.backward(gy.double())
    assert rel_err(y, yr) < 3e-6
    for a, b in zip((h, agg, W1, b1, W2, b2), leaves):
        assert rel_err(a.grad, b.grad) < 5e-6, rel_err(a.grad, b.grad)


@pytest.mark.parametrize("H,emb", [(256, None), (192, 8), (64, 4)])
def test_edge_first_layer_function(H, emb):
    """EdgeFirstLayer: the re-layout of an edge MLP's first Linear (P | Q projections, the two distance columns, the
    bias + edge-type table) and its backward vs autograd over the slicing expression it replaces."""
    from diffsbdd_amd.train_hip import EdgeFirstLayer
    g = torch.Generator().manual_seed(H)
    cols = 2 * H + 2 + (emb or 0)
    w = (torch.randn(H, cols, generator=g) * 0.1).to(dev()).requires_grad_(True)
    b = torch.randn(H, generator=g).to(dev()).requires_grad_(True)
    e = torch.randn(3, emb, generator=g).to(dev()).requires_grad_(True) if emb else None
    outs = EdgeFirstLayer.apply(w, b, e)
    gs = [torch.randn(o.shape, generator=g).to(dev()) for o in outs]
    torch.autograd.backward(outs, gs)
    wd_, bd_ = w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    ed_ = e.detach().double().requires_grad_(True) if emb else None
    ref = (torch.cat((wd_[:, :H], wd_[:, H:2 * H]), 0), wd_[:, 2 * H], wd_[:, 2 * H + 1],
           (bd_[None, :] + ed_ @ wd_[:, 2 * H + 2:].t()) if emb else bd_[None, :].expand(3, H))
    torch.autograd.backward(ref, [x.double() for x in gs])
    for o, r in zip(outs, ref):
        assert rel_err(o, r) < 1e-6
    assert rel_err(w.grad, wd_.grad) < 1e-6 and rel_err(b.grad, bd_.grad) < 1e-6
    if emb:
        assert rel_err(e.grad, ed_.grad) < 1e-6


def _graph_problem(arch, n_lig, n_poc, seed):
    from diffsbdd_amd.train_hip import TrainGraph
    cfg, _ = W.arch_cfg(arch)
    sd = W.random_state_dict(cfg, seed=3)
    m = make_dynamics(cfg, sd)
    xl, xp, t, ml, mp = problem(cfg, n_lig, n_poc, seed)
    x = torch.cat((xl[:, :3], xp[:, :3]), 0).to(dev()).contiguous()
    g = TrainGraph(m, ml.to(dev()), mp.to(dev()), x)
    return cfg, m, g, x


def _edge_ref(pq, x, ed0, row, col, n_lig, wd, wd0, tab, W2, b2, H):
    """a2 = SiLU(SiLU(P[row] + Q[col] + d wd + d0 wd0 + tab[type]) W2^T + b2)   in the tensors' dtype"""
    d = ((x[row] - x[col]) ** 2).sum(1)
    rl, cl = row < n_lig, col < n_lig
    ty = torch.zeros_like(row)
    ty[rl & cl] = 1
    ty[~rl & ~cl] = 2
    z1 = pq[row, :H] + pq[col, H:2 * H] + d[:, None] * wd[None] + ed0[:, None] * wd0[None] + tab[ty]
    return F.silu(F.silu(z1) @ W2.t() + b2), d


@pytest.mark.parametrize("arch,n_lig,n_poc,attention", [
    ("small_cond", [5, 7, 6], [40, 35, 38], True),
    ("small_cond", [9, 0, 4], [30, 50, 0], False),
    ("crossdock_ca_cond", [23, 20, 25, 11], [36, 40, 30, 33], True),
    ("crossdock_fullatom_cond", [23, 18], [286, 250], True),
])
def test_edge_gcl_function_vs_torch_fp64(arch, n_lig, n_poc, attention):
    """EdgeGCL: the aggregate and the gradients w.r.t. pq, x, ed0 and every parameter of the MLP (incl. the attention
    head) against float64 torch autograd of the literal formula; the rows span several 32-edge wave tiles."""
    from diffsbdd_amd.train_hip import EdgeGCL
    cfg, m, g, x = _graph_problem(arch, n_lig, n_poc, seed=5)
    H = cfg["hidden_nf"]
    gen = torch.Generator().manual_seed(11)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(dev())
    pq = rnd(g.N, 2 * H, sc=0.5)
    wd, wd0, tab = rnd(H, sc=0.05), rnd(H, sc=0.05), rnd(3, H, sc=0.3)
    W2, b2 = rnd(H, H, sc=1.0 / H ** 0.5), rnd(H, sc=0.1)
    aw, ab = (rnd(1, H, sc=0.2), rnd(1, sc=0.1)) if attention else (None, None)
    gagg = rnd(g.N, H)
    ed0 = g.ed0[:g.E].clone()
    leaves = [pq, x.clone(), ed0, wd, wd0, tab, W2, b2] + ([aw, ab] if attention else [])
    for v in leaves:
        v.requires_grad_(True)
    args = leaves + ([None, None] if not attention else [])
    agg = EdgeGCL.apply(*args, g, 100.0)
    agg.backward(gagg)
    # float64 reference
    dl = [v.detach().double().requires_grad_(True) for v in leaves]
    row, col = g.erow[:g.E].long(), g.ecol[:g.E].long()
    a2, _ = _edge_ref(dl[0], dl[1], dl[2], row, col, g.n_lig, dl[3], dl[4], dl[5], dl[6], dl[7], H)
    if attention:
        a2 = a2 * torch.sigmoid(a2 @ dl[8].t() + dl[9])
    ref = torch.zeros(g.N, H, dtype=torch.float64, device=dev()).index_add_(0, row, a2) / 100.0
    ref.backward(gagg.double())
    assert int((g.rev[:g.E] < 0).sum()) == 0                     # the radius graph is symmetric
    assert rel_err(agg, ref) < 2e-5, rel_err(agg, ref)
    names = ["pq", "x", "ed0", "wd", "wd0", "tab", "W2", "b2", "att_w", "att_b"]
    for nme, a, b in zip(names, leaves, dl):
        e = rel_err(a.grad, b.grad)
        assert e < 1e-4, (nme, e)


@pytest.mark.parametrize("arch,n_lig,n_poc", [
    ("small_cond", [5, 7, 6], [40, 35, 38]),
    ("small_joint", [6, 3], [30, 41]),
    ("small_variant", [8, 5], [33, 29]),
    ("crossdock_fullatom_cond", [23, 18], [286, 250]),
])
def test_edge_coord_function_vs_torch_fp64(arch, n_lig, n_poc):
    """EdgeCoord: x_out and the gradients w.r.t. pq, x, the sample mean, ed0 and every parameter of the two coordinate
    MLPs against float64 torch autograd of egnn_new.py:96-122 + coord2diff / coord2cross."""
    from diffsbdd_amd.train_hip import EdgeCoord, SampleMean
    cfg, m, g, x = _graph_problem(arch, n_lig, n_poc, seed=7)
    hp = m._hp
    H = cfg["hidden_nf"]
    n_mlp = 1 if hp["reflection_equivariant"] else 2
    gen = torch.Generator().manual_seed(13)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(dev())
    pq = rnd(g.N, 2 * H * n_mlp, sc=0.5)
    groups = []
    for q in range(n_mlp):
        groups += [rnd(H, sc=0.05), rnd(H, sc=0.05), rnd(3, H, sc=0.3), rnd(H, H, sc=1.0 / H ** 0.5), rnd(H, sc=0.1)]
    w3 = rnd(1, H, sc=0.3)
    xx = x.clone()
    ed0 = g.ed0[:g.E].clone()
    gx = rnd(g.N, 3)
    leaves = [pq, xx, ed0] + groups + [w3]
    for v in leaves:
        v.requires_grad_(True)
    mean = SampleMean.apply(xx, g) if n_mlp == 2 else None
    gq = groups + [None] * (5 * (2 - n_mlp))
    x_out = EdgeCoord.apply(pq, xx, mean, ed0, *gq, w3, g, hp)
    x_out.backward(gx)
    # float64 reference
    dl = [v.detach().double().requires_grad_(True) for v in leaves]
    pqd, xd, ed0d = dl[0], dl[1], dl[2]
    w3d = dl[-1]
    row, col = g.erow[:g.E].long(), g.ecol[:g.E].long()
    nb = g.node_batch.long()
    nc, rng = float(hp["norm_constant"]), float(hp["coords_range"])
    diff = xd[row] - xd[col]
    radial = (diff ** 2).sum(1, keepdim=True)
    u = diff / (torch.sqrt(radial + 1e-8) + nc)
    trans = 0
    for q in range(n_mlp):
        wd, wd0, tab, W2, b2 = dl[3 + 5 * q: 8 + 5 * q]
        a2, _ = _edge_ref(pqd[:, 2 * H * q:], xd, ed0d, row, col, g.n_lig, wd, wd0, tab, W2, b2, H)
        phi = a2 @ w3d.t()
        if hp["tanh"]:
            phi = torch.tanh(phi) * rng
        if q == 0:
            trans = u * phi
        else:
            cnt = torch.bincount(nb, minlength=g.batch).clamp(min=1).double()
            mean_d = torch.zeros(g.batch, 3, dtype=torch.float64, device=dev()).index_add_(0, nb, xd) / cnt[:, None]
            a, b = xd[row] - mean_d[nb[row]], xd[col] - mean_d[nb[col]]
            cr = torch.cross(a, b, dim=1)
            trans = trans + cr / (torch.linalg.norm(cr, dim=1, keepdim=True) + nc) * phi
    aggx = torch.zeros(g.N, 3, dtype=torch.float64, device=dev()).index_add_(0, row, trans) / float(hp["normalization_factor"])
    if not hp["update_pocket_coords"]:
        aggx = aggx * (torch.arange(g.N, device=dev()) < g.n_lig)[:, None]
    ref = xd + aggx
    ref.backward(gx.double())
    assert rel_err(x_out, ref) < 2e-5, rel_err(x_out, ref)
    names = ["pq", "x", "ed0"] + [f"{n}_{q}" for q in range(n_mlp) for n in ("wd", "wd0", "tab", "W2", "b2")] + ["w3"]
    for nme, a, b in zip(names, leaves, dl):
        e = rel_err(a.grad, b.grad)
        assert e < 1e-4, (nme, e)


def _grads_vs_oracle(arch, n_lig, n_poc, seed, tol=1e-4, coord_scale=None):
    cfg, _ = W.arch_cfg(arch)
    sd = W.random_state_dict(cfg, seed=1)
    m = make_dynamics(cfg, sd)
    m.train(True)
    xl, xp, t, ml, mp = problem(cfg, n_lig, n_poc, seed)
    gen = torch.Generator().manual_seed(99)
    wl, wp = torch.randn(xl.shape, generator=gen), torch.randn(xp.shape, generator=gen)
    out_l, out_p = m(xl.to(dev()), xp.to(dev()), t.to(dev()), ml.to(dev()), mp.to(dev()))
    assert out_l.requires_grad
    loss = (out_l * wl.to(dev())).sum() + (out_p * wp.to(dev())).sum()
    loss.backward()
    # oracle under autograd on the CPU, on the edge list the HIP builder produced (teacher-forced graph)
    from diffsbdd_amd.train_hip import TrainGraph
    x = torch.cat((xl[:, :3], xp[:, :3]), 0).to(dev()).contiguous()
    edges = TrainGraph(m, ml.to(dev()), mp.to(dev()), x).edges().cpu()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_l, ref_p, _ = eo.dynamics_forward(sdr, cfg, xl, xp, t, ml, mp, edges=edges)
    ref_loss = (ref_l * wl).sum() + (ref_p * wp).sum()
    ref_loss.backward()
    assert (out_l.detach().cpu() - ref_l.detach()).abs().max().item() < 1e-4
    worst = {}
    for pname, p in m.named_parameters():
        g_ref = sdr[pname].grad
        if pname.endswith("coord_mlp.4.weight"):
            twin = pname.replace("coord_mlp", "cross_product_mlp")
            if twin in sdr and sdr[twin].grad is not None:
                g_ref = g_ref + sdr[twin].grad
        assert p.grad is not None and g_ref is not None, pname
        scale = max(g_ref.abs().max().item(), 1e-6)
        worst[pname] = (p.grad.cpu() - g_ref).abs().max().item() / scale
    bad = {k: v for k, v in worst.items() if not v <= tol}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    return max(worst.values()), edges.shape[1]


@pytest.mark.parametrize("arch", ["small_cond", "small_joint", "small_variant"])
def test_dynamics_training_gradients_vs_oracle_small(arch):
    worst, _ = _grads_vs_oracle(arch, [5, 7, 6], [40, 35, 38], seed=21)
    print(arch, "worst relative gradient error", worst)


def test_dynamics_training_gradients_vs_oracle_full_width():
    """crossdock_fullatom_cond (H = 256, 6 blocks) at B = 8: E > 40 k edges, rows spanning many wave tiles, the
    persistent multi-tile loops of the backward kernels; every parameter gradient vs the oracle's autograd at 1e-4."""
    B = 8
    worst, E = _grads_vs_oracle("crossdock_fullatom_cond", [23] * B, [286] * B, seed=22)
    assert E > 40_000, E
    print("full width: E =", E, "worst relative gradient error", worst)


def test_training_gradients_bitwise_reproducible_and_match_torch_path():
    """The backward kernels sum in a fixed order: two runs give identical gradients; and the A/B path of round 3
    (train_path.py, eager torch ops) agrees to rounding."""
    cfg, _ = W.arch_cfg("crossdock_ca_cond")
    sd = W.random_state_dict(cfg, seed=1)
    m = make_dynamics(cfg, sd)
    m.train(True)
    xl, xp, t, ml, mp = problem(cfg, [23, 20, 25, 11], [36, 40, 30, 33], seed=4)
    args = [v.to(dev()) for v in (xl, xp, t, ml, mp)]

    def run():
        m.zero_grad(set_to_none=True)
        o, _ = m(*args)
        (o ** 2).sum().backward()
        return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    a, b = run(), run()
    assert all(torch.equal(a[k], b[k]) for k in a)
    os.environ["DSBDD_TRAIN"] = "torch"
    try:
        c = run()
    finally:
        os.environ.pop("DSBDD_TRAIN")
    assert set(a) == set(c)
    for k in a:
        assert rel_err(a[k], c[k]) < 1e-4, (k, rel_err(a[k], c[k]))


@pytest.mark.parametrize("arch", ["small_cond", "small_joint"])
def test_input_gradients_match_the_torch_path(arch):
    """Gradients w.r.t. the INPUTS (z_t depends on the schedule's parameters when the noise schedule is learned,
    en_diffusion.py:65,378): d loss / d xh_atoms and d xh_residues through the HIP Functions -- the radius-graph
    distances d0 as a differentiable input (EdgeRadial), the coordinate gradients of both edge stages, the encoders --
    against the eager torch path (train_path.py) on the same call."""
    cfg, _ = W.arch_cfg(arch)
    sd = W.random_state_dict(cfg, seed=1)
    m = make_dynamics(cfg, sd)
    m.train(True)
    xl, xp, t, ml, mp = problem(cfg, [5, 7, 6], [40, 35, 38], seed=31, spread=0.6 if arch == "small_joint" else 3.0)
    gen = torch.Generator().manual_seed(5)
    wl, wp = torch.randn(xl.shape, generator=gen).to(dev()), torch.randn(xp.shape, generator=gen).to(dev())

    def run():
        a = xl.to(dev()).clone().requires_grad_(True)
        b = xp.to(dev()).clone().requires_grad_(True)
        o_l, o_p = m(a, b, t.to(dev()), ml.to(dev()), mp.to(dev()))
        ((o_l * wl).sum() + (o_p * wp).sum()).backward()
        return o_l.detach(), a.grad.clone(), b.grad.clone()
    o_h, ga_h, gb_h = run()
    os.environ["DSBDD_TRAIN"] = "torch"
    try:
        o_t, ga_t, gb_t = run()
    finally:
        os.environ.pop("DSBDD_TRAIN")
    assert rel_err(o_h, o_t) < 1e-5
    assert rel_err(ga_h, ga_t) < 1e-4, rel_err(ga_h, ga_t)
    assert rel_err(gb_h, gb_t) < 1e-4, rel_err(gb_h, gb_t)
    # eval mode: inputs that require grad still get their gradients (same kernels, same bits); without them the call is
    # the inference engine and carries no grad_fn (dynamics.py forward docstring)
    m.eval()
    o_e, ga_e, gb_e = run()
    assert torch.equal(ga_e, ga_h) and torch.equal(gb_e, gb_h) and torch.equal(o_e, o_h)
    o_l, o_p = m(xl.to(dev()), xp.to(dev()), t.to(dev()), ml.to(dev()), mp.to(dev()))
    assert o_l.grad_fn is None and not o_l.requires_grad
    assert rel_err(o_l, o_h) < 1e-4


@pytest.mark.parametrize("arch,n_lig,n_poc", [("crossdock_fullatom_cond", [23] * 6, [286] * 6),
                                              ("crossdock_ca_cond", [23, 20, 25, 11], [36, 40, 30, 33]),
                                              ("small_joint", [5, 7, 6], [40, 35, 38])])
def test_side_streams_of_the_backward_keep_the_bits(arch, n_lig, n_poc):
    """Round 6: the network backward runs the weight gradients and the second coordinate MLP's chain on side streams
    (DSBDD_TRAIN_STREAMS, a bit mask read when the per-module handle is created; csrc/engine.hip TrainSide).  Kernels and
    reduction orders are those of the single-stream sequence, so every parameter and input gradient must be IDENTICAL --
    a missing fork / join would show up here as a difference between the masks or between repeats."""
    import copy
    cfg, _ = W.arch_cfg(arch)
    sd = W.random_state_dict(cfg, seed=1)
    xl, xp, t, ml, mp = problem(cfg, n_lig, n_poc, seed=8, spread=0.6 if arch == "small_joint" else 3.0)
    gen = torch.Generator().manual_seed(6)
    wl, wp = torch.randn(xl.shape, generator=gen).to(dev()), torch.randn(xp.shape, generator=gen).to(dev())

    def run(m):
        m.zero_grad(set_to_none=True)
        a = xl.to(dev()).clone().requires_grad_(True)
        b = xp.to(dev()).clone().requires_grad_(True)
        o_l, o_p = m(a, b, t.to(dev()), ml.to(dev()), mp.to(dev()))
        ((o_l * wl).sum() + (o_p * wp).sum()).backward()
        out = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        out["d_xh_atoms"], out["d_xh_residues"] = a.grad.clone(), b.grad.clone()
        return out
    results = {}
    old = os.environ.get("DSBDD_TRAIN_STREAMS")
    try:
        for mask in ("0", "15", "7"):
            os.environ["DSBDD_TRAIN_STREAMS"] = mask
            m = make_dynamics(cfg, copy.deepcopy(sd))          # a new module: a new handle, created under this mask
            m.train(True)
            results[mask] = [run(m) for _ in range(3)]
    finally:
        if old is None:
            os.environ.pop("DSBDD_TRAIN_STREAMS", None)
        else:
            os.environ["DSBDD_TRAIN_STREAMS"] = old
    ref = results["0"][0]
    assert len(ref) > 10
    for mask, runs in results.items():
        for i, r in enumerate(runs):
            assert set(r) == set(ref)
            for k in ref:
                assert torch.equal(r[k], ref[k]), (mask, i, k, (r[k] - ref[k]).abs().max().item())


@pytest.mark.parametrize("arch,n_lig,n_poc", [("crossdock_fullatom_cond", [23] * 6, [286] * 6),
                                              ("crossdock_ca_cond", [23, 20, 25, 11], [36, 40, 30, 33]),
                                              ("small_variant", [5, 7, 6], [40, 35, 38])])
def test_kept_z2_backward_agrees_with_the_recompute_backward(arch, n_lig, n_poc):
    """Round 6: the forward pass of the network path keeps z2 = W2 a1 + b2 of every edge MLP and the backward pass runs the
    element-wise kernels E / EC (csrc/train.h) where rounds 4 - 5 recomputed the H x H layer in kernel A
    (DSBDD_TRAIN_STORE_Z2=0, read when the handle is created).  Same mathematics, different summation order of the row
    sums: every parameter and input gradient agrees to rounding (1e-5 of the gradient's scale; the oracle comparisons
    above hold the 1e-4 bound for the default), and each path repeats itself bit for bit."""
    import copy
    cfg, _ = W.arch_cfg(arch)
    sd = W.random_state_dict(cfg, seed=1)
    xl, xp, t, ml, mp = problem(cfg, n_lig, n_poc, seed=9)
    gen = torch.Generator().manual_seed(7)
    wl, wp = torch.randn(xl.shape, generator=gen).to(dev()), torch.randn(xp.shape, generator=gen).to(dev())

    def run(m):
        m.zero_grad(set_to_none=True)
        a = xl.to(dev()).clone().requires_grad_(True)
        b = xp.to(dev()).clone().requires_grad_(True)
        o_l, o_p = m(a, b, t.to(dev()), ml.to(dev()), mp.to(dev()))
        ((o_l * wl).sum() + (o_p * wp).sum()).backward()
        out = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        out["d_xh_atoms"], out["d_xh_residues"], out["eps_atoms"] = a.grad.clone(), b.grad.clone(), o_l.detach().clone()
        return out
    res = {}
    old = os.environ.get("DSBDD_TRAIN_STORE_Z2")
    try:
        for mode in ("1", "0"):
            os.environ["DSBDD_TRAIN_STORE_Z2"] = mode
            m = make_dynamics(cfg, copy.deepcopy(sd))
            m.train(True)
            res[mode] = [run(m), run(m)]
    finally:
        if old is None:
            os.environ.pop("DSBDD_TRAIN_STORE_Z2", None)
        else:
            os.environ["DSBDD_TRAIN_STORE_Z2"] = old
    for mode, (r0, r1) in res.items():
        for k in r0:
            assert torch.equal(r0[k], r1[k]), (mode, k)
    assert torch.equal(res["1"][0]["eps_atoms"], res["0"][0]["eps_atoms"])        # the forward values do not change
    worst = 0.0
    for k, g1 in res["1"][0].items():
        g0 = res["0"][0][k]
        scale = max(g0.abs().max().item(), 1e-6)
        e = (g1 - g0).abs().max().item() / scale
        worst = max(worst, e)
        assert e < 1e-5, (k, e)
    print(arch, "kept z2 vs recompute: worst relative difference", worst)


@pytest.mark.parametrize("workload,vnode", [("crossdock_fullatom_cond", None), ("crossdock_ca_cond", None),
                                            ("crossdock_fullatom_cond", 3), ("crossdock_ca_cond", "simple")])
def test_loss_terms_on_hip_launches_agree_with_the_torch_terms(workload, vnode):
    """Round 6: in training mode the twelve loss terms of ConditionalDDPM.forward (conditional_model.py:202-330) are
    evaluated by csrc/loss_head.h -- one launch before the network call, one after, one in backward -- instead of ~ 250
    torch launches (DSBDD_LOSS=torch keeps those).  Same t (incl. t = 0: the L0 terms), same keyed noise: every term, the two
    logged means, the normalised batch left in the dictionaries and every parameter gradient of the l2 objective agree to
    rounding (1e-5 of the term's scale; per-sample sums are fixed-order here, atomics in torch)."""
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from train_step_bench import build, loss_of
    from diffsbdd_amd import synthetic as S
    key = "ca" if "ca_" in workload else "fa"
    B = 6
    model, cfg, dd = build(workload, dev())
    model.train(True)
    if vnode == "simple":      # SimpleConditionalDDPM (conditional_model.py:702-746): no COM projection, dof = 3 n
        from diffsbdd_amd.conditional_model import SimpleConditionalDDPM
        model.__class__ = SimpleConditionalDDPM
        vnode = None
    model.vnode_idx = vnode
    rng = np.random.default_rng(3)
    t_fix = torch.tensor(rng.integers(0, dd["timesteps"] + 1, size=(B, 1)), dtype=torch.float32)
    t_fix[0, 0] = 0.0
    t_fix[1, 0] = float(dd["timesteps"])
    model.t_int_source = lambda b: t_fix
    res = {}
    old = os.environ.get("DSBDD_LOSS")
    try:
        for mode in ("torch", "hip"):
            os.environ["DSBDD_LOSS"] = mode
            pocket = S.load_pocket(key, B, dev())
            ligand = S.anchor_ligand(B, 23, cfg["atom_nf"], dev())
            model.seed(77)
            model.zero_grad(set_to_none=True)
            out = model(ligand, pocket, return_info=True)
            loss_of(out[:12]).backward()
            res[mode] = dict(terms=[torch.as_tensor(v).detach().float().cpu() for v in out[:12]],
                             info={k: float(v.detach()) for k, v in out[12].items()},
                             grads={k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None},
                             batch=[ligand['x'].detach().cpu(), ligand['one_hot'].detach().cpu(), pocket['x'].detach().cpu(),
                                    pocket['one_hot'].detach().cpu()])
    finally:
        if old is None:
            os.environ.pop("DSBDD_LOSS", None)
        else:
            os.environ["DSBDD_LOSS"] = old
    names = ("delta_log_px", "error_t_lig", "error_t_pocket", "SNR_weight", "loss_0_x_ligand", "loss_0_x_pocket", "loss_0_h",
             "neg_log_constants", "kl_prior", "log_pN", "t_int", "xh_lig_hat")
    for nme, a, b in zip(names, res["hip"]["terms"], res["torch"]["terms"]):
        assert a.shape == b.shape, (nme, a.shape, b.shape)
        scale = max(1.0, b.abs().max().item())
        assert (a - b).abs().max().item() <= 1e-5 * scale, (nme, (a - b).abs().max().item(), scale)
    assert res["torch"]["terms"][4][0].abs().item() > 0 and res["torch"]["terms"][1][0].item() == 0      # the t = 0 sample: L0_x live, error_t masked
    for k, v in res["torch"]["info"].items():
        assert abs(res["hip"]["info"][k] - v) <= 1e-5 * max(1.0, abs(v)), k
    for a, b in zip(res["hip"]["batch"], res["torch"]["batch"]):
        assert a.shape == b.shape and (a - b).abs().max().item() <= 1e-6
    assert set(res["hip"]["grads"]) == set(res["torch"]["grads"])
    for k, g in res["torch"]["grads"].items():
        scale = max(g.abs().max().item(), 1e-6)
        assert (res["hip"]["grads"][k] - g).abs().max().item() <= 2e-5 * scale, (k, (res["hip"]["grads"][k] - g).abs().max().item(), scale)


def test_edge_capacity_kernel_equals_the_torch_expressions():
    """engine.edge_capacity on GPU masks is one launch (csrc/loss_head.h edge_capacity_kernel) + the one host sync; it must
    return what the torch expressions return for the same masks (they still serve CPU tensors) and reject what they reject."""
    from diffsbdd_amd.engine import edge_capacity
    g = torch.Generator().manual_seed(0)
    for batch in (1, 3, 16, 300):
        nl = torch.randint(0, 40, (batch,), generator=g)
        npk = torch.randint(0, 400, (batch,), generator=g)
        ml = torch.repeat_interleave(torch.arange(batch), nl)
        mp = torch.repeat_interleave(torch.arange(batch), npk)
        assert edge_capacity(ml.to(dev()), mp.to(dev()), batch) == edge_capacity(ml, mp, batch)
    ml = torch.tensor([0, 0, 1, 1, 2]); mp = torch.tensor([0, 1, 1, 2, 2, 2])
    for bad_l, bad_p, b in ((ml.flip(0), mp, 3), (ml, mp.flip(0), 3), (ml, mp, 2), (ml - 1, mp, 3)):
        with pytest.raises(ValueError, match="sorted"):
            edge_capacity(bad_l.to(dev()), bad_p.to(dev()), b)
